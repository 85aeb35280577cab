#!/usr/bin/env python3
"""experiment: insertion rate when the table is cache-resident (second pass over the same reads: no new keys)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jasper_amd import KmerTable, synth
dev = torch.device("cuda", 0)
for gmb, ls in ((0.05, 18), (0.2, 20), (1.0, 22), (4.0, 24), (16.0, 26), (47.0, 29)):
    gen = torch.Generator(device=dev).manual_seed(1)
    g = synth.torch_genome(gen, int(gmb * 1e6), dev, repeat_frac=0)
    nreads = int(gmb * 1e6 * 30 / 150)
    reads = synth.torch_reads_stream(gen, g, nreads)
    # repeat the stream so that one launch has enough work
    rep = max(1, int(300e6 / reads.numel()))
    big = reads.repeat(rep)
    torch.cuda.synchronize()
    t = KmerTable(37, min_slots=1 << ls)
    t.count_bases_device(reads.data_ptr(), reads.numel())     # creates the keys (may grow the table)
    os.environ["JASPER_EXPERIMENT_PIECE"] = str(1 << 31)
    t.count_bases_device(big.data_ptr(), big.numel())         # pure increments, one launch
    ms, n = t.count_timing()
    info = t.info()
    kmers = nreads * rep * (150 - 37 + 1)
    del os.environ["JASPER_EXPERIMENT_PIECE"]
    print("genome %.2f Mb  table 2^%d (%.0f MB) load %.2f: %.1f ms in %d launches -> %.1f Gk/s" %
          (gmb, info["slots"].bit_length() - 1, info["slots"] * 16 / 1e6, info["distinct"] / info["slots"], ms, n, kmers / ms / 1e6), flush=True)
    t.close()
