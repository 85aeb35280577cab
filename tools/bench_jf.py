#!/usr/bin/env python3
"""time the .jf writer / reader at bench scale (GPU box): python tools/bench_jf.py [genome_mb]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jasper_amd import KmerTable, synth

gmb = float(sys.argv[1]) if len(sys.argv) > 1 else 47.0
K = 37
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(2000)
g = synth.torch_genome(gen, int(gmb * 1e6), dev)
nreads = int(gmb * 1e6 * 30 / 150)
gen = torch.Generator(device=dev).manual_seed(2500)
reads = synth.torch_reads_stream(gen, g, nreads, 150, 0.003)
torch.cuda.synchronize()
t = KmerTable(K, min_slots=int(1.25 * nreads * 150 * 2.1 / 10))
t.count_bases_device(reads.data_ptr(), reads.numel())
info = t.info()
path = "/tmp/bench_mer_counts.jf"
t0 = time.perf_counter()
t.write_jf(path, ["count", "-C", "-m", str(K)])
t1 = time.perf_counter()
size = os.path.getsize(path)
print("write_jf: %d k-mers, %.2f GB in %.2f s (%.2f GB/s of file)" % (info["distinct"], size / 1e9, t1 - t0, size / 1e9 / (t1 - t0)), flush=True)
t2 = time.perf_counter()
u = KmerTable.from_jf(path)
t3 = time.perf_counter()
print("from_jf: %.2f s" % (t3 - t2), flush=True)
assert u.histogram() == t.histogram() and u.info()["distinct"] == info["distinct"]
print("reloaded table has the same histogram")
os.remove(path)
