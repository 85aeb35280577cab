#!/usr/bin/env python3
"""the polishing phase of bench.py alone (GPU box): python tools/bench_polish_steps.py [genome_mb] [reps]   env JASPER_POLISH_DEBUG=1"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from jasper_amd import KmerTable, polisher

gmb = float(sys.argv[1]) if len(sys.argv) > 1 else 47.0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
reads, names, seqs, d_chunks, asm_len, bs, nreads = B.build_workload(torch, dev, 0, 1, gmb, 2)
jf_size = int(nreads * 150 * 2.1 / 10)
t = KmerTable(B.K, min_slots=max(1 << 21, int(1.25 * jf_size)))
t.count_bases_device(reads.data_ptr(), reads.numel())
txt, status = polisher.threshold_from_histo_rows(t.histo_rows())
thr = int(txt)
res = None
for r in range(reps):
    res = None          # (a result that is still alive is brought to the host before the table's arenas are reused)
    t0 = time.perf_counter()
    res = t.polish_batch_device(d_chunks[0], d_chunks[1], thr, B.PASSES, fix=True)
    t1 = time.perf_counter()
    print("rep %d: polish wall %.2f ms device %.2f ms, %d records, qv %s, %d segments, %d lookups" % (r, (t1 - t0) * 1e3, res.seconds * 1e3, res.n_records, res.qv, res.segments, res.lookups), flush=True)
