"""CPU, world_size 2 over gloo: the N>1 host path (read sharding, entry exchange, counter all-reduce)."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from jasper_amd import dist as jd
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # every rank holds a different number of (hash hi, lo, count) entries
        n = 5 + 7 * rank
        mine = torch.arange(n * 3, dtype=torch.int64).reshape(n, 3) + 1000 * (rank + 1)
        others = jd.all_gather_entries(mine)
        exp = torch.cat([torch.arange((5 + 7 * r) * 3, dtype=torch.int64).reshape(-1, 3) + 1000 * (r + 1)
                         for r in range(world) if r != rank])
        ok = torch.equal(others, exp)
        # an empty shard must not break the exchange
        empty = jd.all_gather_entries(torch.zeros((0, 3), dtype=torch.int64) if rank == 0 else mine)
        ok = ok and (empty.shape[0] == (12 if rank == 0 else 0))
        tot = jd.all_reduce_ints([rank + 1, 10])
        ok = ok and tot == [sum(range(1, world + 1)), 10 * world]
        lo, hi = jd.shard_range(101, rank, world)
        ok = ok and (hi - lo) in (50, 51)
        q.put((rank, ok, int(others.shape[0])))
    finally:
        dist.destroy_process_group()


def test_entry_exchange_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True, 12), (1, True, 5)]


# ---- dist.shard_tables: the collective protocol, with a dict standing in for the HBM table -----------------------------
class FakeTable:
    """the methods dist.shard_tables calls, over host memory: entries are (key, 0 | count) int64 pairs, owner = key % n"""

    def __init__(self, fail_attach=False, tag=0):
        self.d = {}
        self.slots = 1 << 4
        self.attached = None
        self.fail_attach = fail_attach
        self.tag = tag
        self.handle_epoch = 0
        self.calls = []

    @staticmethod
    def _view(ptr, n):
        import ctypes
        import numpy as np
        return np.ctypeslib.as_array((ctypes.c_int64 * (2 * n)).from_address(ptr)).reshape(n, 2)

    def info(self):
        return {"distinct": len(self.d), "slots": self.slots}

    def export_owner(self, ptr, cap, n):
        groups = [[(k, c) for k, c in sorted(self.d.items()) if k % n == o] for o in range(n)]
        if cap:
            v = self._view(ptr, n * cap)
            for o, g in enumerate(groups):
                for i, (k, c) in enumerate(g[:cap]):
                    v[o * cap + i] = (k, c)
        return [len(g) for g in groups]

    def clear(self):
        self.d = {}

    def sync(self):
        pass

    def reserve(self, min_slots):
        while self.slots < min_slots:
            self.slots *= 2
            self.handle_epoch += 1            # a moved slot array has a new handle

    def fit(self, load):
        want = 16
        while len(self.d) > load * want:
            want *= 2
        if want != self.slots:
            self.slots = want
            self.handle_epoch += 1

    def import_packed(self, ptr, n, mode):
        assert mode == 0
        for k, c in self._view(ptr, n):
            self.d[int(k)] = self.d.get(int(k), 0) + int(c)
        while len(self.d) > 0.5 * self.slots:     # grows like the real table
            self.slots *= 2
            self.handle_epoch += 1

    def import_packed_multi(self, ptrs, counts):
        for p, n in zip(ptrs, counts):
            self.import_packed(p, n, 0)

    def ipc_handle(self):
        return bytes([self.tag, self.handle_epoch % 256]) + bytes(62)

    def attach_ipc(self, handles, me):
        self.calls.append("attach")
        if self.fail_attach:
            raise RuntimeError("no peer access")
        self.attached = [h[:2] for h in handles]

    def detach(self):
        self.attached = None


def _shard_worker(rank, world, port, q, fail_on):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from jasper_amd import dist as jd
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cpu")
        local, shard = FakeTable(), FakeTable(fail_attach=(rank == fail_on), tag=rank + 1)
        out = []
        for step in range(3):
            # rank r holds keys r*5 .. r*5+n-1 (overlapping between ranks), uneven sizes, step 1 overflows the row estimate
            if step == 1 and rank == 1:
                local.d = {2 * k: 3 for k in range(200000)}        # all owned by rank 0: more than the agreed row length
            else:
                n = 40 if rank == 0 else 9
                local.d = {k: (k % 7) + 1 + step for k in range(rank * 5, rank * 5 + n)}
            try:
                got = jd.shard_tables(local, shard, dev)
            except jd.ShardAttachError:
                out.append("attach-failed")
                continue
            out.append((got, dict(shard.d) if len(shard.d) < 100 else len(shard.d), shard.slots, list(shard.attached), list(shard.calls)))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def _run_shard(fail_on):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_shard_worker, args=(r, 2, port, q, fail_on)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_shard_tables_protocol_world2():
    res = _run_shard(fail_on=-1)
    for step in (0, 2):
        exp = {}
        for r in range(2):
            n = 40 if r == 0 else 9
            for k in range(r * 5, r * 5 + n):
                exp[k] = exp.get(k, 0) + (k % 7) + 1 + step
        for r in range(2):
            got, d, slots, attached, calls = res[r][step]
            assert d == {k: c for k, c in exp.items() if k % 2 == r}          # owner r holds exactly its keys, summed
            assert got == sum(1 for rr in range(2) for k in range(rr * 5, rr * 5 + (40 if rr == 0 else 9)) if k % 2 == r)
    # step 1: the row estimate (from the local sizes) was too small for rank 1's table -> exact resend, nothing lost
    assert res[0][1][1] == 200000 and res[1][1][1] == {k: (k % 7) + 2 for k in range(40) if k % 2}
    # one geometry on both owners at every step; re-attached exactly when a slot array moved (steps 0 and 1, not step 2)
    for step in range(3):
        assert res[0][step][2] == res[1][step][2]
        assert res[0][step][3] == res[1][step][3]                                 # both ranks hold the same handle list
    assert res[0][0][4] == ["attach"] and len(res[0][1][4]) == 2 and len(res[0][2][4]) == 2   # nothing moved in step 2: no re-attach


def test_shard_tables_attach_failure_is_collective():
    res = _run_shard(fail_on=1)
    assert res[0] == ["attach-failed"] * 3 and res[1] == ["attach-failed"] * 3      # both ranks raise together, nobody hangs


# ---- dist.count_sharded: the collective protocol of the list exchange, with host memory standing in for HBM ----------------
class FakeExchangeTable(FakeTable):
    """the methods dist.count_sharded calls.  The "reads" are bytes, every byte one record whose key is its value; owner =
    key % n; a send list holds `cap` records, what does not fit is deferred -- like the real lists, only tiny."""

    def __init__(self, geometry=True, fail_scan=False, fail_reserve=False, fail_plan_after=None, **kw):
        super().__init__(**kw)
        self.geometry, self.fail_scan, self.fail_reserve, self.fail_plan_after = geometry, fail_scan, fail_reserve, fail_plan_after
        self.scanned = None
        self.whole = []
        self.plans = []
        self.stage_calls = []    # "s" a sender stage's partition, "i" an owner's insert, in the order they were made

    @staticmethod
    def _arr(ptr, n, ctype):
        import ctypes
        import numpy as np
        return np.ctypeslib.as_array((ctype * n).from_address(ptr))

    def _cap(self, piece_max, records_max, n):
        return max(2, (records_max or piece_max) // n)            # the mean fill of the fullest sender: skewed keys overflow

    def reserve(self, slots):
        if self.fail_reserve:
            raise RuntimeError("HIP out of memory (on purpose)")         # what torch.cuda.OutOfMemoryError is: a RuntimeError of ONE rank
        return super().reserve(slots)

    def exchange_plan(self, piece_max, n, records_max=0):
        if self.fail_plan_after is not None and len(self.plans) >= self.fail_plan_after:
            raise RuntimeError("plan failed on purpose")
        if not self.geometry:
            return None
        cap = self._cap(piece_max, records_max, n)
        self.plans.append((piece_max, records_max))
        return dict(records_per_owner=cap, counts_per_owner=1, deferred_cap=1024, slice_cap=cap, p1=0, p2=0, p2_owner=0, region_bits=0, slices=1)

    def exchange_scan(self, ptr, n, pos, end, piece_max, nown, d_deferred, dcap):
        import ctypes
        if self.fail_scan:
            raise RuntimeError("scan failed on purpose")
        assert end - pos <= piece_max
        self.scanned = [int(b) for b in self._arr(ptr, n, ctypes.c_uint8)[pos:end]] if n else []
        self._arr(d_deferred, 8, ctypes.c_int64)[:] = 0
        return len(self.scanned)

    def exchange_partition(self, piece_max, records_max, nown, d_send, d_send_cnt, d_deferred, dcap):
        import ctypes
        cap = self._cap(piece_max, records_max, nown)
        send = self._arr(d_send, nown * cap, ctypes.c_int64).reshape(nown, cap)
        cnt = self._arr(d_send_cnt, nown, ctypes.c_int32)
        dfr = self._arr(d_deferred, 8 + 3 * dcap, ctypes.c_int64)
        cnt[:] = 0
        self.stage_calls.append("s")
        for key in self.scanned:
            o = key % nown
            if cnt[o] < cap:
                send[o, cnt[o]] = key
                cnt[o] += 1
            else:
                i = int(dfr[0])
                if i < dcap:
                    dfr[8 + 3 * i: 8 + 3 * i + 3] = (0, key, 1)
                dfr[0] = i + 1
        self.scanned = None

    def exchange_insert(self, d_recv, d_recv_cnt, piece_max, records_max, nown, me, d_all=0, n_all=0, whole_input=False, slice_cap=0, count_bits=0):
        assert not slice_cap and not count_bits            # (this table's plan says p2 = 0: no bits for counts, nothing is deduplicated)
        import ctypes
        cap = self._cap(piece_max, records_max, nown)
        recv = self._arr(d_recv, nown * cap, ctypes.c_int64).reshape(nown, cap)
        cnt = self._arr(d_recv_cnt, nown, ctypes.c_int32)
        for src in range(nown):
            for key in recv[src, :cnt[src]]:
                assert int(key) % nown == me
                self.d[int(key)] = self.d.get(int(key), 0) + 1
        if n_all:
            ent = self._arr(d_all, 3 * n_all, ctypes.c_int64).reshape(n_all, 3)
            for _, key, inc in ent:
                if int(key) % nown == me:
                    self.d[int(key)] = self.d.get(int(key), 0) + int(inc)
        self.whole.append(bool(whole_input))
        self.stage_calls.append("i")


def _reads_of(rank, step):
    import numpy as np
    rng = np.random.default_rng(100 * step + rank)
    if step == 1 and rank == 1:
        return np.zeros(0, dtype=np.uint8)                       # a rank with nothing to count
    n = 400 if rank == 0 else 150
    a = rng.integers(0, 40, n).astype(np.uint8)
    if step == 2:
        a[: n // 2] = 6                                          # one heavy key: its list overflows, records travel deferred
    return a


def _count_worker(rank, world, port, q, mode, pipeline="1"):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), JASPER_AMD_EXCHANGE_PIPELINE=pipeline)
    import torch
    import torch.distributed as dist
    from jasper_amd import dist as jd
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cpu")
        shard = FakeExchangeTable(geometry=(mode != "nogeom" or rank == 0), fail_scan=(mode == "fail" and rank == 1),
                                  fail_reserve=(mode == "fail_reserve" and rank == 1), fail_plan_after=(2 if mode == "fail_plan" and rank == 0 else None), tag=rank + 1)
        out = []
        for step in range(4):
            reads = torch.from_numpy(_reads_of(rank, step % 3).copy())
            try:
                info = jd.count_sharded(shard, reads.data_ptr() if reads.numel() else 0, reads.numel(), dev, clear=(step < 3),
                                        piece_limit=(None if step != 0 else 128))
            except jd.CollectiveCountError as e:          # (only what EVERY rank raises: another exception fails this worker)
                out.append("raised: " + str(e)[:40])
                continue
            if info is None:
                out.append(None)
                continue
            out.append((dict(shard.d), info["rounds"], info["deferred"], list(shard.whole), shard.slots, list(shard.attached), "".join(shard.stage_calls)))
            shard.whole.clear()
            del shard.stage_calls[:]
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def _run_count(mode, world=2, pipeline="1"):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_count_worker, args=(r, world, port, q, mode, pipeline)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_count_sharded_protocol_world2():
    res = _run_count("ok")
    carried = {}
    for step in range(4):
        exp = dict(carried) if step == 3 else {}                  # step 3 does not clear: it adds to what step 2 left
        for r in range(2):
            for key in _reads_of(r, step % 3):
                exp[int(key)] = exp.get(int(key), 0) + 1
        carried = exp
        for r in range(2):
            d, rounds, deferred, whole, slots, attached, calls = res[r][step]
            assert d == {k: c for k, c in exp.items() if k % 2 == r}          # owner r holds exactly its keys, summed over ranks and rounds
        assert res[0][step][1] == res[1][step][1] and res[0][step][2] == res[1][step][2]
        assert res[0][step][4] == res[1][step][4] and res[0][step][5] == res[1][step][5]      # one geometry, same handle list
    assert res[0][0][1] == 4                     # step 0: 400 and 150 bytes in pieces of 128 -> 4 rounds, rank 1 idle in the last two
    assert res[0][0][3] == [False] * 4           # ... and no round is "the whole input"
    assert res[0][1][1] == 1 and res[0][1][3] == [True]          # step 1: one round although rank 1 has nothing
    assert res[0][2][2] > 0                      # step 2: the heavy key overflowed its list and arrived as deferred entries
    assert res[0][3][3] == [False]               # step 3: one round, but the shard was not empty
    # the rounds are a pipeline: a round's lists are inserted only after the NEXT round's sender stage (its all_to_all is under way
    # meanwhile); the last round's insert closes the call
    assert res[0][0][6] == res[1][0][6] == "ssisisii"
    assert res[0][1][6] == "si"


def test_count_sharded_stages_one_after_the_other_give_the_same_shards():
    """JASPER_AMD_EXCHANGE_PIPELINE=0 (what bench.py falls back to when the pipeline's pattern fails the transport self-test): every
    round sender -> all_to_all -> insert before the next one starts; same shards, rounds and deferred records as the pipeline"""
    a, b = _run_count("ok"), _run_count("ok", pipeline="0")
    for r in range(2):
        for step in range(4):
            assert a[r][step][:6] == b[r][step][:6]
    assert b[0][0][6] == "sisisisi"


def test_count_sharded_protocol_world8():
    """the same protocol with the 8 ranks of one node (gloo, host memory): every owner ends up with exactly its keys, summed over
    all ranks and rounds; rounds, deferred records and the geometry are the same on every rank"""
    W = 8
    res = _run_count("ok", world=W)
    carried = {}
    for step in range(4):
        exp = dict(carried) if step == 3 else {}
        for r in range(W):
            for key in _reads_of(r, step % 3):
                exp[int(key)] = exp.get(int(key), 0) + 1
        carried = exp
        for r in range(W):
            assert res[r][step][0] == {k: c for k, c in exp.items() if k % W == r}
            assert res[r][step][1:3] == res[0][step][1:3] and res[r][step][4:6] == res[0][step][4:6]
    assert res[0][0][1] == 4 and res[0][2][2] > 0          # four rounds in step 0; the heavy key of step 2 travelled as deferred records


def test_count_sharded_failures_are_collective():
    res = _run_count("fail")
    assert all(isinstance(x, str) and x.startswith("raised") for r in range(2) for x in res[r])      # both ranks raise together, nobody hangs
    res = _run_count("nogeom")
    assert res[0] == [None] * 4 and res[1] == [None] * 4          # one rank without a geometry: nobody consumes anything
    # what only ONE rank can run into -- a device allocation while growing its shard, a library call between two reductions --
    # is agreed on before anybody leaves: both ranks raise CollectiveCountError at the same point, nobody hangs or falls back alone
    res = _run_count("fail_reserve")
    assert all(isinstance(x, str) and x.startswith("raised") for r in range(2) for x in res[r])
    res = _run_count("fail_plan")
    assert all(isinstance(x, str) and x.startswith("raised") for r in range(2) for x in res[r])
