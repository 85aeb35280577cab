#!/usr/bin/env python3
"""the counting phase of bench.py alone, repeated on one table (GPU box): stage times of the path taken.
   python tools/bench_count_steps.py [genome_mb] [reps]      env JASPER_COUNT_DEBUG=2"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jasper_amd import KmerTable, synth

gmb = float(sys.argv[1]) if len(sys.argv) > 1 else 47.0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
K = int(os.environ.get("K", "37"))
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(2000)
g = synth.torch_genome(gen, int(gmb * 1e6), dev)
nreads = int(gmb * 1e6 * 30 / 150)
gen = torch.Generator(device=dev).manual_seed(2500)
reads = synth.torch_reads_stream(gen, g, nreads)
torch.cuda.synchronize()
kmers = nreads * (150 - K + 1)
jf_size = int(nreads * 150 * 2.1 / 10)
t = KmerTable(K, min_slots=max(1 << 21, int(1.25 * jf_size)))
for r in range(reps):
    if os.environ.get("FRESH_TABLE") and r:      # (experiment builds whose pieces abandon themselves: a table takes the partition passes once)
        t.close()
        t = KmerTable(K, min_slots=max(1 << 21, int(1.25 * jf_size)))
    t.clear()
    t.sync()
    t0 = time.perf_counter()
    t.count_bases_device(reads.data_ptr(), reads.numel())
    t.sync()
    t1 = time.perf_counter()
    ms, n = t.count_timing()
    st, pl = t.count_stages()
    path = t.count_path()
    info = t.info()
    h = t.histogram()
    print("rep %d: path %d wall %.2f ms kernel %.2f ms -> %.1f Gk/s, %.0f GB/s algorithmic (%.3f of 8 TB/s); stages %s; distinct %d occ %d histo[1..3] %s"
          % (r, path, (t1 - t0) * 1e3, ms, kmers / ms / 1e6, 33 * kmers / ms / 1e6, 33 * kmers / ms / 1e6 / 8000,
             " ".join("%s=%.2f" % (nm.replace("_kernel", ""), v) for nm, v in zip(KmerTable.STAGE_NAMES[path], st)), info["distinct"], info["occurrences"], h[1:4]), flush=True)
    t.count_timing_reset() if hasattr(t, "count_timing_reset") else None
