#!/usr/bin/env python3
"""time the GPU-side steps of the owner-sharded exchange (dist.shard_tables) on ONE GPU: `world` local tables = the ranks'
read shards of the world x genome; owner 0's role is played in full (export grouped by owner, add every rank's segment 0
into the shard, histogram of the shard) and the polishing of one rank's chunks runs through a sharded view whose other
owners also live on this GPU (so the xGMI hop is NOT in these numbers).  python tools/bench_shard.py [genome_mb] [world]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jasper_amd import KmerTable, synth, polisher

gmb = float(sys.argv[1]) if len(sys.argv) > 1 else 47.0
world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
K = 37
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(2000)
G = int(gmb * 1e6)
g = synth.torch_genome(gen, G * world, dev)                         # the N x genome of the weak-scaling bench
nreads = int(gmb * 1e6 * 30 / 150)                                 # per rank
slots = int(1.25 * nreads * 150 * 2.1 / 10)


def timed(label, fn, tabs=()):
    for t in tabs: t.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); out = fn()
    for t in tabs: t.sync()
    torch.cuda.synchronize()
    print("%-60s %8.2f ms" % (label, (time.perf_counter() - t0) * 1e3), flush=True)
    return out


# every rank's local table, exported by owner; only the segments are kept (the local tables would not all fit for N=8)
segs = [[None] * world for _ in range(world)]      # segs[src][owner]
local = KmerTable(K, min_slots=slots)
for r in range(world):
    gen = torch.Generator(device=dev).manual_seed(2500 + r)
    reads = synth.torch_reads_stream(gen, g, nreads, 150, 0.003)
    torch.cuda.synchronize()
    local.clear()
    timed("rank %d: count its read shard" % r, lambda: local.count_bases_device(reads.data_ptr(), reads.numel()), [local])
    del reads
    d = local.info()["distinct"]
    cap = int(d / world * 1.05) + (1 << 16)
    send = torch.empty((world, cap, 2), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    counts = timed("rank %d: export %d keys grouped by owner (2^%d slots)" % (r, d, local.info()["slots"].bit_length() - 1),
                   lambda: local.export_owner(send.data_ptr(), cap, world), [local])
    for o in range(world):
        segs[r][o] = send[o, :counts[o]].clone()
    del send
local.close()
torch.cuda.synchronize()

shards = []
for o in range(world):
    t = KmerTable(K, min_slots=1 << 21)
    incoming = sum(int(segs[r][o].shape[0]) for r in range(world))
    t.reserve(2 * incoming)
    t.import_packed_multi([segs[r][o].data_ptr() for r in range(world)], [segs[r][o].shape[0] for r in range(world)])
    t.fit(0.5)
    shards.append(t)
print("owner tables: %s keys in 2^%d slots" % ([t.info()["distinct"] for t in shards], shards[0].info()["slots"].bit_length() - 1), flush=True)
S = shards[0]
for rep in range(2):
    timed("owner 0: clear + add %d incoming entries" % sum(int(segs[r][0].shape[0]) for r in range(world)),
          lambda: (S.clear(), [S.import_packed(segs[r][0].data_ptr(), segs[r][0].shape[0], 0) for r in range(world)]), [S])
for rep in range(2):
    timed("owner 0: clear + add the same in ONE sweep (import_packed_multi)",
          lambda: (S.clear(), S.import_packed_multi([segs[r][0].data_ptr() for r in range(world)], [segs[r][0].shape[0] for r in range(world)])), [S])
for rep in range(2):
    h = timed("owner 0: histogram of its shard", lambda: S.histogram(), [S])
acc = [0] * 10002
for t in shards:
    acc = [a + b for a, b in zip(acc, t.histogram())]
thr, status = polisher.threshold_from_histo_rows([(m, acc[m]) for m in range(1, 10002) if acc[m]])
thr = int(thr)
for o, t in enumerate(shards):
    t.attach_tables(shards, o)
# rank 0's chunks: the first genome unit with assembly errors
import numpy as np
asm = torch.from_numpy(synth.make_assembly(np.random.default_rng(7), g[:G].cpu().numpy())).to(dev)
L = int(asm.numel())
bs = int(G / 16 * 0.9)
offs = list(range(0, L, bs)) + [L]
torch.cuda.synchronize()
res = None
for rep in range(4):
    res = None        # (a live result of the previous call is flushed to the host first)
    res = timed("polish %d chunks through the sharded view (thr %d)" % (len(offs) - 1, thr), lambda: S.polish_batch_device(asm, offs, thr, 2), [S])
print("qv", res.qv, "fix records", res.n_records)
