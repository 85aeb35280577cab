#!/usr/bin/env python3
"""counting-only timing with the per-kernel stage split (GPU box): python tools/exp_count.py [genome_mb] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jasper_amd import KmerTable, synth

gmb = float(sys.argv[1]) if len(sys.argv) > 1 else 47.0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
K = int(os.environ.get("K", "37"))
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(2000)
g = synth.torch_genome(gen, int(gmb * 1e6), dev)
nreads = int(gmb * 1e6 * 30 / 150)
gen = torch.Generator(device=dev).manual_seed(2500)
reads = synth.torch_reads_stream(gen, g, nreads, 150, 0.003)
torch.cuda.synchronize()
kmers = nreads * (150 - K + 1)
t = KmerTable(K, min_slots=max(1 << 21, int(1.25 * nreads * 150 * 2.1 / 10)))
for r in range(reps):
    t.clear()
    t.sync()
    t1 = time.perf_counter()
    t.count_bases_device(reads.data_ptr(), reads.numel())
    t.sync()
    t2 = time.perf_counter()
    ms, n = t.count_timing()
    st, pl = t.count_stages()
    info = t.info()
    print("rep %d: wall %.2f ms kernel %.2f ms (%d launches) stages %s -> %.2f Gk/s; distinct %d occ %d slots 2^%d"
          % (r, (t2 - t1) * 1e3, ms, n, ["%.2f" % x for x in st], kmers / (t2 - t1) / 1e9, info["distinct"], info["occurrences"],
             info["slots"].bit_length() - 1), flush=True)
h = t.histogram()
print("histo[1..6]", h[1:7], "sum", sum(h))
