"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed ("nccl" = RCCL over xGMI on ROCm).

What shards and what is exchanged:
  * reads are independent units and counts form a commutative monoid (sum by key)  ->  every rank counts its own
    shard of the reads into its own HBM table; the per-GPU tables are then merged KEY-WISE.  Per-GPU open-addressed
    tables do not share slot positions, so a dense all-reduce over raw tables would be wrong; instead every rank
    exports its (mixed hash, count) entries, the entry lists are all-gathered (equal-size padded buffers, one
    collective that uses every xGMI link), and each rank re-inserts the other ranks' entries into its own table.
    This is the role of `jellyfish merge` (JF::jellyfish/merge_files.cc:44-176) in the reference's tool set.
  * chunk records are independent (the reference's own xargs -P parallelism, src/jasper.sh:212)  ->  greedy
    size-balanced assignment of chunks to ranks; nothing is exchanged except two integers per rank for the QV.

The functions that talk to torch.distributed work on plain tensors so that the same code runs under gloo on CPU
(tests/test_dist_gloo.py, world_size 2) and under RCCL on the GPUs.
"""
import os


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_range(n_items, rank, world):
    """contiguous, balanced [lo, hi) of n_items for this rank"""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def assign_chunks(lengths, world):
    """greedy longest-first balancing of chunk records over ranks; returns rank of every chunk (deterministic)"""
    load = [0] * world
    owner = [0] * len(lengths)
    for i in sorted(range(len(lengths)), key=lambda i: (-lengths[i], i)):
        r = min(range(world), key=lambda r: (load[r], r))
        owner[i] = r
        load[r] += lengths[i]
    return owner


def all_gather_entries(entries, group=None):
    """entries: int64 tensor [n, 3] (mixed hash hi, lo, count) of THIS rank, on the device the backend works with.

    Returns one int64 tensor [m, 3] holding the entries of all OTHER ranks (order: by rank).  One all_gather of
    the sizes, one all_gather_into_tensor of buffers padded to the largest size.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = torch.tensor([entries.shape[0]], dtype=torch.int64, device=entries.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    send = torch.zeros((mx, 3), dtype=torch.int64, device=entries.device)
    if entries.shape[0]:
        send[: entries.shape[0]] = entries
    recv = torch.empty((world, mx, 3), dtype=torch.int64, device=entries.device)
    try:
        dist.all_gather_into_tensor(recv.view(world * mx, 3), send, group=group)
    except (RuntimeError, NotImplementedError):
        parts = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(parts, send, group=group)
        recv = torch.stack(parts)
    others = [recv[r, : sizes[r]] for r in range(world) if r != rank and sizes[r]]
    if not others:
        return torch.zeros((0, 3), dtype=torch.int64, device=entries.device)
    return torch.cat(others, dim=0).contiguous()


def merge_tables(table, device):
    """key-wise sum of the per-GPU tables: afterwards every rank's table holds the counts of ALL reads"""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    info = table.info()
    mine = torch.empty((max(info["distinct"], 1), 3), dtype=torch.int64, device=device)
    n = table.export_to(mine.data_ptr(), mine.shape[0])
    torch.cuda.synchronize(device)
    others = all_gather_entries(mine[:n])
    torch.cuda.synchronize(device)
    if others.shape[0]:
        table.import_device(others.data_ptr(), others.shape[0])
    return int(others.shape[0])


def all_reduce_ints(values, device=None):
    """sum a short list of python ints over ranks (QV counters, k-mer totals)"""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return list(values)
    t = torch.tensor(list(values), dtype=torch.int64, device=device)
    dist.all_reduce(t)
    return [int(x) for x in t.tolist()]
