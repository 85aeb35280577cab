#!/bin/bash
# GPU box: SQ counters of the counting kernels (two separate --pmc passes), output under gpurun_out/$1
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-pmc}
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq1 -- python3 tools/bench_count_steps.py 47 2 > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/sq2 -- python3 tools/bench_count_steps.py 47 2 > $OUT/sq2.log 2>&1
python3 tools/pmc_summary.py $OUT/sq1 > $OUT/sq_counters.txt 2>&1
python3 tools/pmc_summary.py $OUT/sq2 >> $OUT/sq_counters.txt 2>&1
rm -rf $OUT/sq1 $OUT/sq2
