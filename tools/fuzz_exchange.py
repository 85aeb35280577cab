#!/usr/bin/env python3
"""the GPU fuzz (HIP path vs oracle, tests/test_gpu_fuzz.py) with the COUNTING done the multi-GPU way on the one GPU: the read
file goes through the read feed (jasper_read_feed_*), every batch is cut into 2..5 ranges -- one per virtual rank, the cuts
wherever they fall, so k-mers span them -- each range is scanned and partitioned into region lists by key owner
(jasper_count_exchange_scan / _partition), the blocks are regrouped as an all_to_all would, and every owner inserts what it
received into its shard (jasper_count_exchange_insert).  Histogram, lookups and polishing then run through the shards.
Tables without an exchange geometry (wide remainders) are counted whole and split by owner as in tools/fuzz_shard.py.
   python tools/fuzz_exchange.py SEED0 N [logfile]"""
import os, sys, tempfile, pathlib, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def exchanged_class(KmerTable):
    import torch
    from fuzz_shard import sharded_class
    Base = sharded_class(KmerTable)

    class Exchanged(Base):
        """count_files fills n owner shards through the list exchange; everything else reads through them (Base)"""
        taken = [0, 0, 0]         # cases counted by the exchange / counted whole and split / batches whose lists were deduplicated

        def count_files(self, paths):
            info = KmerTable.info(self)
            size = sum(os.path.getsize(p) for p in paths)
            n = 2 + (size + self.k) % 4
            if KmerTable.exchange_plan(self, 1 << 24, n) is None:
                Exchanged.taken[1] += 1
                return KmerTable.count_files(self, paths)
            Exchanged.taken[0] += 1
            shards = [KmerTable(self.k, min_slots=info["slots"]) for _ in range(n)]
            feeder = KmerTable(self.k, min_slots=1 << 10)
            feeder.feed_start([(p, 0, -1) for p in paths])
            try:
                while True:
                    ptr, nb = feeder.feed_next()
                    if nb == 0:
                        break
                    cuts = [nb * r // n for r in range(n + 1)]
                    piece = max(b - a for a, b in zip(cuts, cuts[1:]))
                    dcap = shards[0].exchange_plan(piece, n)["deferred_cap"]
                    dfr = [torch.empty(8 + 3 * dcap, dtype=torch.int64, device="cuda") for _ in range(n)]
                    torch.cuda.synchronize()
                    found = [shards[r].exchange_scan(ptr, nb, cuts[r], cuts[r + 1], piece, n, dfr[r].data_ptr(), dcap) for r in range(n)]
                    feeder.feed_release()
                    rmax = max(max(found), 1)
                    plan = shards[0].exchange_plan(piece, n, rmax)
                    send = [torch.empty((n, plan["records_per_owner"]), dtype=torch.int64, device="cuda") for _ in range(n)]
                    cnt = [torch.empty((n, plan["counts_per_owner"]), dtype=torch.int32, device="cuda") for _ in range(n)]
                    torch.cuda.synchronize()
                    for r in range(n):
                        shards[r].exchange_partition(piece, rmax, n, send[r].data_ptr(), cnt[r].data_ptr(), dfr[r].data_ptr(), dcap)
                        shards[r].sync()
                    nd = [int(d[0].item()) for d in dfr]
                    assert max(nd) <= dcap
                    d_all = torch.cat([d[8:8 + 3 * m] for d, m in zip(dfr, nd)]).contiguous() if sum(nd) else None
                    slice_cap, cbits = 0, 0
                    if size % 3 and plan["p2"] >= 1:      # (two cases in three) the lists deduplicated by their senders, packed
                        dd = [shards[r].exchange_dedupe(piece, rmax, n, send[r].data_ptr(), cnt[r].data_ptr()) for r in range(n)]
                        cbits, slice_cap = dd[0][1], max(max(d[0] for d in dd), 1)
                        lists = n * plan["counts_per_owner"]
                        send = [t.view(lists, plan["slice_cap"])[:, :slice_cap].contiguous().view(n, -1) for t in send]
                        Exchanged.taken[2] += 1
                    for o in range(n):
                        recv = torch.stack([send[r][o] for r in range(n)]).contiguous()
                        rcnt = torch.stack([cnt[r][o] for r in range(n)]).contiguous()
                        torch.cuda.synchronize()
                        shards[o].exchange_insert(recv.data_ptr(), rcnt.data_ptr(), piece, rmax, n, o, d_all.data_ptr() if d_all is not None else 0, sum(nd),
                                                  slice_cap=slice_cap, count_bits=cbits)
            finally:
                feeder.close()
            # one geometry for all owners (a shard that had to grow), then every read goes through the sharded view
            slots = max(KmerTable.info(t)["slots"] for t in shards)
            for t in shards:
                t.reserve(slots)
            for o, t in enumerate(shards):
                t.attach_tables(shards, o)
            self._shards = shards

    return Exchanged


if __name__ == "__main__":
    import test_gpu_fuzz as T
    from jasper_amd import KmerTable, polisher
    from oracle import oracle as O
    import make_golden as G
    import fuzz_vs_reference as F
    seed0, n = int(sys.argv[1]), int(sys.argv[2])
    log = open(sys.argv[3], "a") if len(sys.argv) > 3 else sys.stdout
    S = exchanged_class(KmerTable)
    tmp = pathlib.Path(tempfile.mkdtemp(prefix="fuzzxchg_"))
    t0 = time.time()
    bad = 0
    for i, seed in enumerate(range(seed0, seed0 + n)):
        try:
            T._one(seed, S, polisher, O, G, F, tmp)
        except AssertionError as e:
            bad += 1
            log.write("seed %d FAILED: %s\n" % (seed, str(e)[:300]))
        for f in tmp.iterdir():
            f.unlink()
        if (i + 1) % 100 == 0:
            log.write("%d cases, %d failing, %.0f s (%d by exchange, %d split, %d batches deduplicated)\n" % (i + 1, bad, time.time() - t0, S.taken[0], S.taken[1], S.taken[2]))
            log.flush()
    log.write("done (counting by list exchange): seeds %d..%d, %d failing; %d cases by exchange (%d batches with deduplicated lists), %d counted whole and split\n" % (seed0, seed0 + n - 1, bad, S.taken[0], S.taken[2], S.taken[1]))
