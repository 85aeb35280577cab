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
