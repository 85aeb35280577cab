#!/usr/bin/env python3
"""jasper_count_bases (bases in HOST memory, the boundary a SWIG-style binding would call) on bench.py's reads: PCIe-inclusive
counting rate, against the same bases already resident in HBM.   python tools/bench_host_count.py [genome_mb]
   env JASPER_COUNT_HOST_STREAM=1: the round-1 form (64-MiB pieces straight into the direct kernel)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from jasper_amd import KmerTable

gmb = float(sys.argv[1]) if len(sys.argv) > 1 else 47.0
dev = torch.device("cuda", 0)
reads = bench.build_workload(torch, dev, 0, 1, gmb, 2)[0]
host = reads.cpu().numpy().tobytes()
n = len(host)
nreads = n // (bench.READ_LEN + 1)
kmers = nreads * (bench.READ_LEN - bench.K + 1)
slots = max(1 << 21, int(1.25 * nreads * bench.READ_LEN * 2.1 / 10))
ref = KmerTable(bench.K, min_slots=slots)
ref.count_bases_device(reads.data_ptr(), n)
want = (ref.info()["distinct"], ref.info()["occurrences"], ref.histogram()[:6])
t = KmerTable(bench.K, min_slots=slots)
for rep in range(3):
    t.clear()
    t.sync()
    t0 = time.perf_counter()
    t.count_bases(host)
    t.sync()
    dt = time.perf_counter() - t0
    got = (t.info()["distinct"], t.info()["occurrences"], t.histogram()[:6])
    print("rep %d: %d bases from host memory in %.1f ms -> %.1f Gk-mers/s, %.1f GB/s of bases over PCIe; equal to the HBM-resident count: %s"
          % (rep, n, dt * 1e3, kmers / dt / 1e9, n / dt / 1e9, got == want), flush=True)
    assert got == want
