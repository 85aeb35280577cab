#!/usr/bin/env python3
"""time the local (GPU-side) steps of the multi-GPU table merge on ONE GPU: two tables = two ranks' read shards of the
same genome; rank 0's role is played in full (export to owners, add what the other rank sends, export own range, set the
other owner's final range).  python tools/bench_merge.py [genome_mb] [world]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jasper_amd import KmerTable, synth

gmb = float(sys.argv[1]) if len(sys.argv) > 1 else 47.0
world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
K = 37
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(2000)
g = synth.torch_genome(gen, int(gmb * 1e6 * world), dev)          # the N x genome of the weak-scaling bench
nreads = int(gmb * 1e6 * 30 / 150)                                 # per rank
slots = int(1.25 * nreads * world * 150 * 2.1 / 10)
tabs = []
for r in range(2):                                                 # only two of the `world` shards are materialised
    gen = torch.Generator(device=dev).manual_seed(2500 + r)
    reads = synth.torch_reads_stream(gen, g, nreads, 150, 0.003)
    torch.cuda.synchronize()
    t = KmerTable(K, min_slots=slots)
    t.count_bases_device(reads.data_ptr(), reads.numel())
    tabs.append(t)
    del reads
A, B = tabs
ia, ib = A.info(), B.info()
print("world %d: rank tables hold %d / %d distinct keys in 2^%d slots" % (world, ia["distinct"], ib["distinct"], ia["slots"].bit_length() - 1), flush=True)

def timed(label, fn):
    A.sync(); B.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = fn(); A.sync(); B.sync(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    print("%-46s %8.2f ms" % (label, dt), flush=True)
    return out

cap = int(ia["distinct"] / world * 1.25) + (1 << 16)
send = torch.empty((world, cap, 2), dtype=torch.int64, device=dev)
sendB = torch.empty((cap, 2), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
cnt = timed("export to all owners (one table pass)", lambda: [0 if o == 0 else A.export_packed(send[o].data_ptr(), cap, o, world) for o in range(world)])
nb = timed("(other rank) export for owner 0", lambda: B.export_packed(sendB.data_ptr(), cap, 0, world))
timed("add %d received entries" % nb, lambda: A.import_packed(sendB.data_ptr(), nb, 0))
n_mine = timed("size of my final range", lambda: A.export_packed(0, 0, 0, world))
mine = torch.empty((n_mine, 2), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
timed("export my final range (%d)" % n_mine, lambda: A.export_packed(mine.data_ptr(), n_mine, 0, world))
# the other owner's final range, as it would arrive by all_gather: B plays owner 1 (its own partial range stands in for the final one)
n1 = B.export_packed(0, 0, 1, world)
other = torch.empty((n1, 2), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
B.export_packed(other.data_ptr(), n1, 1, world)
timed("set %d entries of another owner's final range" % n1, lambda: A.import_packed(other.data_ptr(), n1, 1))
print("after: %d distinct" % A.info()["distinct"])
