#!/usr/bin/env python3
"""GPU box: a batch of very many tiny chunk records (a fragmented assembly) against the oracle: python tools/many_chunks_probe.py [n_chunks]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from jasper_amd import KmerTable, synth, polisher
from oracle import oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
k, thre, passes = 37, 3, 2
rng = np.random.default_rng(77)
genome = synth.make_genome(rng, 3_000_000)
reads = synth.make_reads_stream(rng, genome, 30, 150, 0.003).tobytes()
asm = synth.make_assembly(rng, genome, err=2e-3).tobytes().decode()
t = KmerTable(k, min_slots=1 << 22)
t.count_bases(reads)
db = O.OracleDB(k)
db.count_bases(reads)
starts = rng.integers(0, len(asm) - 500, n)
lens = rng.integers(20, 400, n)
seqs = [asm[a:a + l] for a, l in zip(starts, lens)]
names = ["c%d:0" % i for i in range(n)]
t0 = time.perf_counter()
fixed, rows, qv, res = polisher.polish_batch(t, names, seqs, thre, passes)
t1 = time.perf_counter()
print("GPU path: %d chunks, %.1f Mb, %.2f s wall, %.1f ms device, %d segments, qv %s" % (n, sum(lens) / 1e6, t1 - t0, res.seconds * 1e3, res.segments, qv), flush=True)
fixed_o, rows_o, qv_o, _ = db.polish_batch(names, seqs, thre, passes)
print("oracle: %.2f s" % (time.perf_counter() - t1))
ok = fixed == fixed_o and qv == qv_o and all(polisher.fix_csv_text(rows[i]) == "Contig Base_coord Original Mutation\r\n" + rows_o[i] for i in range(passes))
print("equal to the oracle:", ok)
sys.exit(0 if ok else 1)
