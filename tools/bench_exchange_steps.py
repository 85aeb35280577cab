#!/usr/bin/env python3
"""The kernels of the multi-GPU counting exchange (dist.count_sharded) timed on ONE GPU, one virtual rank after the other:
every rank's reads (bench.py's workload for a world of W ranks) are partitioned into region lists grouped by owner, the blocks
meant for owner 0 are kept, and owner 0 inserts them into its shard.  Prints the stage times of a sender (part1, part2 by owner)
and of an owner (region_insert), and checks the shard against the keys owner 0 gets from a plain table of all reads'
k-mers when W is small enough for that table to fit.
   python tools/bench_exchange_steps.py [W] [genome_mb] [reps]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from jasper_amd import KmerTable

W = int(sys.argv[1]) if len(sys.argv) > 1 else 2
gmb = float(sys.argv[2]) if len(sys.argv) > 2 else 47.0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
K = bench.K
dev = torch.device("cuda", 0)
nreads = int(gmb * 1e6 * bench.COVERAGE / bench.READ_LEN)
jf_size = int(nreads * bench.READ_LEN * 2.1 / 10)
shard = KmerTable(K, min_slots=max(1 << 21, int(1.25 * jf_size)))
sender = KmerTable(K, min_slots=max(1 << 21, int(1.25 * jf_size)))      # (a second table of the same geometry plays the senders: own stage timers)
n = nreads * (bench.READ_LEN + 1)
kmers = nreads * (bench.READ_LEN - K + 1)
plan = shard.exchange_plan(n, W, kmers)
print("W %d, %d bases per rank, plan %s" % (W, n, plan), flush=True)
assert plan is not None
nrec, ncnt, dcap = plan["records_per_owner"], plan["counts_per_owner"], plan["deferred_cap"]
send = torch.empty((W, nrec), dtype=torch.int64, device=dev)
cnt = torch.empty((W, ncnt), dtype=torch.int32, device=dev)
dfr = torch.empty(8 + 3 * dcap, dtype=torch.int64, device=dev)
recv = torch.empty((W, nrec), dtype=torch.int64, device=dev)
rcnt = torch.empty((W, ncnt), dtype=torch.int32, device=dev)
from jasper_amd import dist as jdist
DEDUPE = (os.environ["DEDUPE"] != "0") if "DEDUPE" in os.environ else jdist.dedupe_pays(W)      # (the rule of dist.count_sharded: dedupe for 2..4 ranks)
slice_cap, cbits = 0, 0
prev = [0.0] * 8
for rep in range(reps):
    part = []
    for r in range(W):
        reads = bench.build_workload(torch, dev, r, W, gmb, 2)[0]
        assert reads.numel() == n
        torch.cuda.synchronize()
        for _ in range(2 if (rep == 0 and r == 0) else 1):      # (first launch: warm-up)
            assert sender.exchange_scan(reads.data_ptr(), n, 0, n, n, W, dfr.data_ptr(), dcap) == kmers
            sender.exchange_partition(n, kmers, W, send.data_ptr(), cnt.data_ptr(), dfr.data_ptr(), dcap)
            sender.sync()
        ndef = int(dfr[0].item())
        if DEDUPE:
            before = int(cnt.to(torch.int64).sum().item())
            t0 = time.perf_counter()
            dd = sender.exchange_dedupe(n, kmers, W, send.data_ptr(), cnt.data_ptr())
            t_dd = (time.perf_counter() - t0) * 1e3
            assert dd is not None
            after = int(cnt.to(torch.int64).sum().item())
            if rep == 0:
                slice_cap, cbits = max(slice_cap, dd[0]), dd[1]          # (the fullest list of any sender sizes what everybody ships)
            if r == W - 1:
                print("   sender %d: dedupe %.2f ms (wall, incl. the wait): %d -> %d records (%.2fx), fullest list %d of cap %d" % (r, t_dd, before, after, before / max(after, 1), dd[0], plan["slice_cap"]), flush=True)
        recv[r].copy_(send[0])
        rcnt[r].copy_(cnt[0])
        torch.cuda.synchronize()
        del reads
        fill = cnt.to(torch.float64)
        part.append((float(fill.mean().item()), int(cnt.max().item())))
    shard.clear()
    shard.sync()
    t0 = time.perf_counter()
    if DEDUPE and rep > 0:       # (from the second repetition on the fullest list is known: pack what owner 0 received as the all_to_all would deliver it)
        packed = recv.view(W * ncnt, plan["slice_cap"])[:, :slice_cap].contiguous()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        shard.exchange_insert(packed.data_ptr(), rcnt.data_ptr(), n, kmers, W, 0, 0, 0, whole_input=True, slice_cap=slice_cap, count_bits=cbits)
    else:
        shard.exchange_insert(recv.data_ptr(), rcnt.data_ptr(), n, kmers, W, 0, 0, 0, whole_input=True, slice_cap=0, count_bits=cbits if DEDUPE else 0)
    shard.sync()
    t1 = time.perf_counter()
    acc, _ = shard.count_stages()                      # (accumulated over the calls on a table that never scans: per-call = difference)
    acc = list(acc) + [0.0] * (8 - len(acc))
    st = [a - b for a, b in zip(acc, prev)]
    prev = acc
    info = shard.info()
    owner_ms = st[2] + st[3] + st[4]
    used = int(rcnt.to(torch.int64).sum().item())
    print("rep %d (last sender deferred %d): owner 0 received %d records (%.3f of one rank's k-mers) in %d x %d slices of cap %d (mean fill %.0f, fullest %d): "
          "region_insert %.2f (+%.2f) deferred %.2f ms (wall %.2f); shard distinct %d in 2^%d slots"
          % (rep, ndef, used, used / kmers, W, ncnt, plan["slice_cap"], part[0][0], max(p[1] for p in part), st[2], st[3], st[4], (t1 - t0) * 1e3,
             info["distinct"], info["slots"].bit_length() - 1), flush=True)
# a sender's stage times: the last partition call on `sender` followed by an (empty-handed) insert would mix tables; read the events through one more full cycle
reads = bench.build_workload(torch, dev, 0, W, gmb, 2)[0]
torch.cuda.synchronize()
best = None
for rep in range(max(reps, 3)):
    sender.clear()
    sender.exchange_scan(reads.data_ptr(), n, 0, n, n, W, dfr.data_ptr(), dcap)
    sender.exchange_partition(n, kmers, W, send.data_ptr(), cnt.data_ptr(), dfr.data_ptr(), dcap)
    sender.sync()
    if DEDUPE:
        sender.exchange_dedupe(n, kmers, W, send.data_ptr(), cnt.data_ptr())
    for r in range(W):
        recv[r].copy_(send[0]) if r == 0 else rcnt[r].zero_()
    rcnt[0].copy_(cnt[0])
    torch.cuda.synchronize()
    sender.exchange_insert(recv.data_ptr(), rcnt.data_ptr(), n, kmers, W, 0, 0, 0, whole_input=False, slice_cap=0, count_bits=cbits if DEDUPE else 0)
    st, _ = sender.count_stages()                      # (the scan resets the stage times of its table)
    print("sender rep %d: part1 %.2f ms, part2 by owner%s %.2f ms" % (rep, st[0], " + dedupe" if DEDUPE else "", st[1]), flush=True)
    if best is None or st[0] + st[1] < best[0] + best[1]:
        best = st
st = best
wire = (W - 1) * ((ncnt * slice_cap if slice_cap else nrec) * 8 + ncnt * 4)
print("sender: part1 %.2f ms, part2 by owner (+ dedupe) %.2f ms; wire per rank %.2f GB as shipped (%.2f GB of raw records); sum sender + owner kernels %.2f ms -> %.1f Gk-mers/s per GPU"
      % (st[0], st[1], wire / 1e9, 8.0 * kmers * (W - 1) / W / 1e9, st[0] + st[1] + 0, kmers / ((st[0] + st[1]) * 1e-3) / 1e9), flush=True)

# what a rank's counting step costs with these kernels, by the bytes-per-link model of DESIGN.md 7 (one xGMI link per peer, all of them busy
# in an all_to_all; NOTHING here has run over xGMI): the stages one after the other, and as the three-stage pipeline of dist.count_sharded
# (the lists of round r travel while the sender kernels of round r + 1 and the owner's insert of round r - 1 run; R = 3 rounds for an input
# of >= 1 G bases that is resident in HBM; ~1 ms of launches and agreements per additional round)
link = jdist.XGMI_LINK_GB_S * 1e9
t_wire = wire / ((W - 1) * link) * 1e3
kern = st[0] + st[1] + owner_ms
R = 3
print("model, W = %d, dedupe %s: kernels %.1f ms (sender %.1f + owner %.1f), wire %.2f GB over %d links = %.1f ms -> one stage after the other %.1f ms, "
      "pipelined (%d rounds) %.1f ms per rank (one GPU alone: part1 + part2f + region_insert ~11.1 ms)"
      % (W, "on" if DEDUPE else "off", kern, st[0] + st[1], owner_ms, wire / 1e9, W - 1, t_wire, kern + t_wire, R, kern + t_wire / R + (R - 1) * 1.0), flush=True)
