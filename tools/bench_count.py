#!/usr/bin/env python3
"""micro-benchmark of the counting kernel alone (GPU box): python tools/bench_count.py [genome_mb] [log2_slots] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jasper_amd import KmerTable, synth

gmb = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
ls = int(sys.argv[2]) if len(sys.argv) > 2 else 28
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
K = int(os.environ.get("K", "37"))
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(1)
g = synth.torch_genome(gen, int(gmb * 1e6), dev)
nreads = int(gmb * 1e6 * 30 / 150)
reads = synth.torch_reads_stream(gen, g, nreads)
torch.cuda.synchronize()
kmers = nreads * (150 - K + 1)
for r in range(reps):
    t0 = time.perf_counter()
    t = KmerTable(K, min_slots=1 << ls)
    t.sync()
    t1 = time.perf_counter()
    t.count_bases_device(reads.data_ptr(), reads.numel())
    t.sync()
    t2 = time.perf_counter()
    ms, n = t.count_timing()
    info = t.info()
    t.close()
    t3 = time.perf_counter()
    print("rep %d: create %.1f ms, count wall %.1f ms, kernel %.1f ms in %d launches -> %.2f Gk/s kernel, %.1f GB/s algorithmic; distinct %d slots 2^%d load %.2f; destroy %.1f ms"
          % (r, (t1 - t0) * 1e3, (t2 - t1) * 1e3, ms, n, kmers / ms / 1e6, 33 * kmers / ms / 1e6, info["distinct"],
             info["slots"].bit_length() - 1, info["distinct"] / info["slots"], (t3 - t2) * 1e3), flush=True)
