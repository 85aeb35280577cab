"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed ("nccl" = RCCL over xGMI on ROCm).

What shards and what is exchanged:
  * reads are independent units and counts form a commutative monoid (sum by key)  ->  every rank counts its own
    shard of the reads into its own HBM table; the per-GPU tables are then summed KEY-WISE.  Per-GPU open-addressed
    tables do not share slot positions, so a dense all-reduce over raw tables would be wrong.  Two ways to do it:
      shard_tables  (default for N>1)  every key has an OWNER GPU (a hash of the key); one all_to_all sends each rank's
                    (mixed hash, count) entries to their owners, who add them up.  The sum stays key-sharded: the
                    polishing kernels read a key from its owner's HBM -- their own or, IPC-mapped, a peer's over xGMI.
                    Cost per GPU independent of the number of GPUs; the table of configs[3]/[4] fits.
      merge_tables  the same reduce-scatter followed by an all-gather: every GPU ends up with the whole merged table
                    (what `jellyfish merge`, JF::jellyfish/merge_files.cc:44-176, would leave on disk).  Cost per GPU
                    grows with the number of GPUs; kept for runs where the peers' memory cannot be mapped.
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


def _fastq_record_start(buf):
    """offset in buf of the first line that starts a 4-line FASTQ record, judged on two consecutive records ('@' / bases /
    '+' / as many quality symbols as bases, twice): a quality line may start with '@' as well, a header cannot be told from
    it on its own.  None if the window holds no such place."""
    starts = []
    i = buf.find(b"\n")
    while i >= 0 and len(starts) < 4096:
        starts.append(i + 1)
        i = buf.find(b"\n", i + 1)
    for a in range(len(starts) - 8):
        L = [buf[starts[a + j]:starts[a + j + 1] - 1] for j in range(8)]
        if (L[0][:1] == b"@" and L[2][:1] == b"+" and L[4][:1] == b"@" and L[6][:1] == b"+" and len(L[1]) == len(L[3]) and len(L[5]) == len(L[7])
                and L[1][:1] not in (b"@", b"+", b"") and L[5][:1] not in (b"@", b"+", b"")):
            return starts[a]
    return None


def plan_read_shards(paths, world):
    """Which bytes of the read files each rank counts: shards[rank] = [(path, begin, end), ...] (end -1 = to the end).

    The reference counts ONE stream, `zcat -f file1 file2 ...` (src/jasper.sh:177); counts are sums over reads, so the
    stream may be cut at any record boundary.  Plain files whose first bytes agree ('@' or '>') and which end in a newline
    are each cut into `world` byte ranges, every cut moved forward to the next record start.  Anything else is not cut:
    gzip files go whole to one rank each (largest first), and input that only makes sense as one stream (formats that
    differ between files, a file without a final newline, multi-line FASTQ where no 4-line record start is found) goes
    whole to rank 0.  Deterministic: every rank computes the same plan."""
    import gzip
    import stat as _stat
    if any(not _stat.S_ISREG(os.stat(p).st_mode) for p in paths):     # a pipe can be read once, by one reader: not looked at here
        return [[(p, 0, -1) for p in paths]] + [[] for _ in range(world - 1)]
    info = []
    for p in paths:
        with open(p, "rb") as f:
            magic = f.read(2)
            size = os.fstat(f.fileno()).st_size
            gz = magic == b"\x1f\x8b"
            last = b"\n"
            if not gz and size:
                f.seek(size - 1)
                last = f.read(1)
        if gz:
            with gzip.open(p, "rb") as g:
                first = g.read(1)
        else:
            first = magic[:1]
        info.append((p, size, gz, first, last))
    everything_to_rank0 = [[(p, 0, -1) for p in paths]] + [[] for _ in range(world - 1)]
    firsts = {i[3] for i in info if i[1]}
    if len(firsts) > 1 or not firsts <= {b"@", b">"} or any(i[4] != b"\n" for i in info[:-1]):
        return everything_to_rank0
    shards = [[] for _ in range(world)]
    whole = [i for i in info if i[2]]
    load = [0] * world
    for p, size, gz, first, last in info:
        if gz or not size:
            continue
        cuts = [0]
        with open(p, "rb") as f:
            for r in range(1, world):
                target = max(size * r // world, cuts[-1])
                pos, win = None, 1 << 20
                while pos is None and target < size:
                    f.seek(target)
                    buf = f.read(win)
                    if first == b">":
                        j = buf.find(b"\n>")
                        pos = target + j + 1 if j >= 0 else None
                    else:
                        j = _fastq_record_start(buf)
                        pos = target + j if j is not None else None
                    if pos is None:
                        if target + len(buf) >= size:
                            break
                        if win >= (64 << 20):
                            return everything_to_rank0        # no 4-line record start in 64 MB: not a file to cut blindly
                        win *= 4
                cuts.append(pos if pos is not None else size)
        cuts.append(size)
        for r in range(world):
            if cuts[r + 1] > cuts[r]:
                shards[r].append((p, cuts[r], cuts[r + 1]))
                load[r] += cuts[r + 1] - cuts[r]
    for p, size, gz, first, last in sorted(whole, key=lambda i: (-i[1], i[0])):
        r = min(range(world), key=lambda r: (load[r], r))
        shards[r].append((p, 0, -1))
        load[r] += 4 * size
    return shards


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


def _sync(device):
    import torch
    if device.type == "cuda":
        torch.cuda.synchronize(device)


def _all_to_all_rows(send, group=None):
    """send[dst] goes to rank dst; returns recv with recv[src] = what rank src sent to this rank.
    RCCL: one all_to_all_single (point-to-point over every xGMI link at once). gloo has no all_to_all for device
    tensors, so the CPU/rehearsal path gathers everything and picks its column -- same result, more traffic."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if dist.get_backend(group) == "nccl":
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv.view(-1), send.view(-1), group=group)
        return recv
    parts = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(parts, send, group=group)
    return torch.stack([parts[src][rank] for src in range(world)])


def _all_to_all_rows_async(send, group=None):
    """_all_to_all_rows without waiting: returns (get, wait) -- wait() blocks the host until the rows have arrived, get() then gives
    recv with recv[src] = what rank src sent to this rank.  RCCL: all_to_all_single(async_op=True) on the communicator's own stream;
    gloo (rehearsal): an asynchronous all_gather, the column picked afterwards."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if dist.get_backend(group) == "nccl":
        recv = torch.empty_like(send)
        work = dist.all_to_all_single(recv.view(-1), send.view(-1), group=group, async_op=True)

        def wait():
            work.wait()                                  # (the current torch stream waits for the collective ...)
            torch.cuda.current_stream(send.device).synchronize()      # (... and the host for that stream: the library works on a stream of its own)
        return (lambda: recv), wait
    parts = [torch.empty_like(send) for _ in range(world)]
    work = dist.all_gather(parts, send, group=group, async_op=True)
    return (lambda: torch.stack([parts[src][rank] for src in range(world)])), (lambda: work.wait())


_CONTROL_GROUPS = {}


def _control_group(group=None):
    """a gloo group over the ranks of `group` for the small agreements of a pipelined exchange (host tensors): a reduction of four
    integers must not queue behind gigabytes of lists on the data communicator.  Made once per data group; every rank makes it at
    the same point of the program (its first pipelined count_sharded)."""
    import torch.distributed as dist
    key = id(group) if group is not None else None
    if key not in _CONTROL_GROUPS:
        if dist.get_backend(group) == "gloo":
            _CONTROL_GROUPS[key] = group                 # (the rehearsal's transport already is one)
        else:
            ranks = dist.get_process_group_ranks(group) if group is not None else None
            _CONTROL_GROUPS[key] = dist.new_group(ranks=ranks, backend="gloo")
    return _CONTROL_GROUPS[key]


def merge_tables(table, device, group=None):
    """key-wise sum of the per-GPU tables; afterwards every rank's table holds the counts of ALL reads.

    Owner-partitioned (reduce-scatter + all-gather by key range) instead of all-gathering whole tables:
      1. ranks agree on one table geometry (all_reduce MAX of the slot count, local rehash if needed);
      2. the key space is cut into `world` slot ranges; every rank sends the entries it holds of range o to rank o
         (all_to_all of 16-byte packed entries) and rank o adds them to its own counts -> range o is final on rank o;
      3. every rank publishes its final range (all_gather) and the others SET those counts in their tables.
    Per rank this moves about (1 + (world-1)/world) x its own entries plus the merged table once, instead of
    (world-1) x all entries.  Returns the number of entries received.
    """
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    # 1. common geometry
    slots = torch.tensor([table.info()["slots"]], dtype=torch.int64, device=device)
    dist.all_reduce(slots, op=dist.ReduceOp.MAX, group=group)
    table.reserve(int(slots.item()))
    # 2. what I hold of every owner's range: ONE pass over the table (each owner's keys sit in their own slot range).
    #    Row length from an estimate (my distinct keys / world + 25 %), agreed by all ranks; the rare overflow falls back
    #    to exact sizes.  libjasper_hip works on its own HIP stream: torch memory must be idle (no pending fill / no
    #    pending work of a previous owner of the cached block) before it is handed over, hence empty() + synchronize,
    #    never zeros()
    est = torch.tensor([int(table.info()["distinct"] / world * 1.25) + (1 << 16)], dtype=torch.int64, device=device)
    dist.all_reduce(est, op=dist.ReduceOp.MAX, group=group)
    mx = int(est.item())
    send = torch.empty((world, mx, 2), dtype=torch.int64, device=device)
    _sync(device)
    mine_counts = [0 if o == rank else table.export_packed(send[o].data_ptr(), mx, o, world) for o in range(world)]
    counts = torch.tensor(mine_counts, dtype=torch.int64, device=device)
    allc = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(allc, counts, group=group)
    allc = torch.stack(allc).cpu()                     # allc[src][dst]
    if int(allc.max().item()) > mx:                    # (every rank sees the same matrix and takes the same branch)
        mx = int(allc.max().item())
        del send
        send = torch.empty((world, mx, 2), dtype=torch.int64, device=device)
        _sync(device)
        for o in range(world):
            if o != rank and int(allc[rank][o]):
                table.export_packed(send[o].data_ptr(), mx, o, world)
    _sync(device)
    recv = _all_to_all_rows(send, group)
    _sync(device)
    del send
    received = 0
    for src in range(world):
        n = int(allc[src][rank])
        if src != rank and n:
            table.import_packed(recv[src].data_ptr(), n, 0)        # add: my range becomes final
            received += n
    del recv
    # 3. publish my final range, take the others'
    n_mine = table.export_packed(0, 0, rank, world)
    nm = torch.tensor([n_mine], dtype=torch.int64, device=device)
    alln = [torch.zeros_like(nm) for _ in range(world)]
    dist.all_gather(alln, nm, group=group)
    alln = [int(x.item()) for x in alln]
    mx2 = max(max(alln), 1)
    mine = torch.empty((mx2, 2), dtype=torch.int64, device=device)
    _sync(device)
    table.export_packed(mine.data_ptr(), mx2, rank, world)
    _sync(device)
    full = torch.empty((world, mx2, 2), dtype=torch.int64, device=device)
    try:
        dist.all_gather_into_tensor(full.view(world * mx2, 2), mine, group=group)
    except (RuntimeError, NotImplementedError):
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        full = torch.stack(parts)
    _sync(device)
    for src in range(world):
        if src != rank and alln[src]:
            table.import_packed(full[src].data_ptr(), alln[src], 1)  # set: the owner's final counts replace my partial ones
            received += alln[src]
    return received


def _ipc_probe_ok(shard, handles, rank, seconds=None):
    """Mapping a peer's memory is a call into the driver stack that has been seen never to return (an allocation of exactly
    2 GiB on ROCm 7.2).  A rank stuck there would hang the whole job, so the same calls are first made by a throw-away
    process with a time limit; only if that comes back clean does the rank itself attach.  JASPER_AMD_IPC_PROBE=0 skips it."""
    import subprocess
    import sys
    if os.environ.get("JASPER_AMD_IPC_PROBE", "1") in ("0", "no", "false") or not hasattr(shard, "device"):
        return True
    seconds = seconds or float(os.environ.get("JASPER_AMD_IPC_PROBE_SECONDS", "60"))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    # Popen with bounded waits: if the helper is stuck in an uninterruptible driver call, SIGKILL does not reap it and an
    # unbounded wait() here would be exactly the hang the probe exists to avoid -- the zombie is abandoned instead.
    import tempfile
    errf = tempfile.TemporaryFile()
    p = subprocess.Popen([sys.executable, "-m", "jasper_amd._ipc_probe", str(shard.device), str(rank), b"".join(bytes(h) for h in handles).hex()],
                         env=env, stdin=subprocess.DEVNULL, stdout=subprocess.DEVNULL, stderr=errf)
    why = ""
    try:
        rc = p.wait(timeout=seconds)
        ok = rc == 0
        if not ok:
            why = "exit %d" % rc
    except subprocess.TimeoutExpired:
        ok = False
        why = "no answer within %.0f s" % seconds
        p.kill()
        try:
            p.wait(timeout=5)
        except subprocess.TimeoutExpired:
            why += " (helper not reaped)"
    if not ok:
        try:
            errf.seek(0)
            tail = errf.read()[-400:].decode("utf-8", "replace").strip()
            sys.stderr.write("jasper_amd: IPC probe of rank %d failed: %s%s\n" % (rank, why, (" -- " + tail) if tail else ""))
        except Exception:
            pass
    errf.close()
    return ok


def _job_processes(token):
    """pids of the processes that carry `token` (a path made by mkdtemp for one job) as a whole command-line argument"""
    me = os.getpid()
    found = []
    for name in os.listdir("/proc"):
        if not name.isdigit() or int(name) == me:
            continue
        try:
            with open("/proc/%s/cmdline" % name, "rb") as f:
                args = f.read().split(b"\0")
        except OSError:
            continue
        if token.encode() in args:
            found.append(int(name))
    return found


def _kill_job_processes(token, seconds=10.0):
    """SIGKILL to exactly the processes of _job_processes(token); returns how many were still there after `seconds`"""
    import signal
    import time
    t_end = time.monotonic() + seconds
    while True:
        pids = _job_processes(token)
        if not pids:
            return 0
        for pid in pids:
            try:
                os.kill(pid, signal.SIGKILL)
            except OSError:
                pass
        if time.monotonic() > t_end:
            return len(pids)
        time.sleep(0.2)


def transport_selftest(world, backend="nccl", one_gpu=False, seconds=240, mb=64):
    """First contact with the transport in throw-away processes (jasper_amd/_selftest.py): `world` ranks, one per GPU, set up a
    process group and run one all_to_all_single of `mb` megabytes, an all_reduce, an all_gather and a barrier -- as a CHILD job
    under a time limit.  A collective that never returns then costs a killed child (its own process group, nothing else) and a
    fall-back to the gloo transport, not a hung job.  Call it before this process touches the GPU; returns a dict for the logs:
    {"ok", "seconds", "ms": {...}, "error"}."""
    import json
    import signal
    import socket
    import subprocess
    import sys
    import tempfile
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = tempfile.mkdtemp(prefix="jasper_selftest_")
    out = os.path.join(d, "result.json")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
           "-m", "jasper_amd._selftest", "--out", out, "--backend", backend, "--mb", str(mb)] + (["--one-gpu"] if one_gpu else [])
    # (the child job makes its own rendezvous: nothing of the parent's, which may itself be a rank of a running job)
    env = {k: v for k, v in os.environ.items() if not (k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE",
                                                             "MASTER_ADDR", "MASTER_PORT", "GROUP_WORLD_SIZE", "ROLE_NAME") or k.startswith("TORCHELASTIC_"))}
    env["PYTHONPATH"] = root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    res = {"ok": False, "world": world, "backend": backend, "ms": {}, "error": None}
    t0 = time.perf_counter()
    try:
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, start_new_session=True)
        try:
            _, errtxt = p.communicate(timeout=seconds)
            files = [out] + [out + ".rank%d" % r for r in range(1, world)]
            parts = []
            for fn in files:
                try:
                    parts.append(json.load(open(fn)))
                except (OSError, ValueError):
                    parts.append(None)
            if parts[0]:
                res.update(parts[0])
            res["ok"] = p.returncode == 0 and all(q and q.get("ok") for q in parts)
            res["pipeline_ok"] = bool(res["ok"] and all(q and q.get("pipeline_ok") for q in parts))
            if not res["ok"] and not res.get("error"):
                bad = [q["error"] for q in parts if q and q.get("error")]
                res["error"] = bad[0] if bad else "self-test exit code %s: %s" % (p.returncode, (errtxt or "")[-300:])
        except subprocess.TimeoutExpired:
            # the launcher puts every rank into a session of its own: killing the launcher's process group alone would leave
            # the ranks behind, on the GPUs.  Ask the launcher to end them (it forwards SIGTERM), then kill it, then kill
            # whatever still carries this job's result path on its command line (the path is unique to this call)
            try:
                p.send_signal(signal.SIGTERM)
                p.communicate(timeout=10)
            except (OSError, subprocess.TimeoutExpired):
                pass
            try:
                os.killpg(p.pid, signal.SIGKILL)            # exactly the process group started above
            except OSError:
                pass
            p.communicate()
            left = _kill_job_processes(out)
            res["error"] = "no answer within %d s: killed" % seconds + (" (%d rank processes did not go away)" % left if left else "")
    except OSError as e:
        res["error"] = "could not start the self-test: %r" % (e,)
    res["seconds"] = round(time.perf_counter() - t0, 2)
    import shutil
    shutil.rmtree(d, ignore_errors=True)
    return res


class CollectiveCountError(RuntimeError):
    """count_sharded failed and EVERY rank of the group raises this at the same point of the protocol (the failure of one
    rank -- a reader error, a device allocation, a list overflow -- is agreed on with a reduction before anybody leaves), so
    the callers may fall back together.  Any other exception out of count_sharded is one rank's alone: it must end that rank,
    never be answered with a fallback that the other ranks do not take."""


class ShardAttachError(RuntimeError):
    """the owners' slot arrays could not be mapped into this process (no IPC / no peer access between the GPUs)"""


def shard_tables(local, shard, device, group=None):
    """key-wise sum of the per-GPU tables WITHOUT replicating the result: owner o keeps the keys with owner_of(hash) == o
    in `shard`, and lookups through `shard` (polishing, Table.lookup) read the owner's HBM directly -- their own, or a
    peer's over xGMI (IPC-mapped).  Per rank: one pass over `local` (export grouped by owner), ONE all_to_all of 16-byte
    entries (the reduce-scatter half of merge_tables), add what arrived -- nothing is all-gathered and no rank ever holds
    more than 1/world of the keys, so the cost per GPU does not grow with the number of GPUs and the table of configs[3]/[4]
    fits.  `shard` keeps its size from call to call (it only ever grows), so repeated calls do not rehash.

    Collective: every rank calls it with its own `local` (counts of its read shard) and `shard` (same k).  Afterwards
    `local` is unchanged.  Returns the number of entries this rank received (its own included).
    """
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        raise RuntimeError("shard_tables needs an initialised process group of more than one rank")
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    # 1. my entries grouped by owner: one pass over the local table.  Row length from an estimate agreed by all ranks
    #    (owners are a hash of the key: rows are even), exact sizes on the rare overflow.  libjasper_hip works on its own
    #    HIP stream: torch memory must be idle before it is handed over, hence empty() + synchronize, never zeros()
    est = torch.tensor([int(local.info()["distinct"] / world * 1.05) + (1 << 16)], dtype=torch.int64, device=device)
    dist.all_reduce(est, op=dist.ReduceOp.MAX, group=group)
    mx = int(est.item())
    send = torch.empty((world, mx, 2), dtype=torch.int64, device=device)
    _sync(device)
    counts = local.export_owner(send.data_ptr(), mx, world)
    ct = torch.tensor(counts, dtype=torch.int64, device=device)
    allc = [torch.zeros_like(ct) for _ in range(world)]
    dist.all_gather(allc, ct, group=group)
    allc = torch.stack(allc).cpu()                     # allc[src][dst]
    if int(allc.max().item()) > mx:                    # (every rank sees the same matrix and takes the same branch)
        mx = int(allc.max().item())
        del send
        send = torch.empty((world, mx, 2), dtype=torch.int64, device=device)
        _sync(device)
        local.export_owner(send.data_ptr(), mx, world)
    _sync(device)                                      # (device-wide: every rank has also finished reading the old shards)
    recv = _all_to_all_rows(send, group)
    _sync(device)
    del send
    # 2. my shard: empty it (after the exchange -- no peer still reads it), add what arrived
    shard.clear()
    incoming = int(allc[:, rank].sum().item())
    first = not getattr(shard, "_fitted", False)
    if first:
        shard.reserve(2 * incoming)                    # a guess that never has to grow (sources share keys: generous)
    srcs = [src for src in range(world) if int(allc[src][rank])]
    if srcs:                                           # all lists in one sweep over the shard (they share its slot order)
        shard.import_packed_multi([recv[src].data_ptr() for src in srcs], [int(allc[src][rank]) for src in srcs])
    del recv
    if first:                                          # ... cut to size once; later calls reuse the slot array as it is
        shard.fit(0.5)
        shard._fitted = True
    _attach_shards(shard, device, group)
    return incoming


def _attach_shards(shard, device, group=None):
    """the owners' shards are written: one geometry for all, every rank maps every owner's slot array (again, if any moved),
    and nobody reads before everybody has finished writing.  Collective."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    # 3. one geometry for all owners, then (re)attach if any slot array moved
    slots = torch.tensor([shard.info()["slots"]], dtype=torch.int64, device=device)
    dist.all_reduce(slots, op=dist.ReduceOp.MAX, group=group)
    shard.reserve(int(slots.item()))
    shard.sync()
    ok, why = 1, ""
    try:
        handle = shard.ipc_handle()
    except RuntimeError as e:                          # (decided collectively below: a rank must not leave the others waiting)
        handle, ok, why = bytes(64), 0, str(e)
    sig = (handle, shard.info()["slots"])
    changed = torch.tensor([0 if ok and getattr(shard, "_attached_sig", None) == sig else 1], dtype=torch.int64, device=device)
    dist.all_reduce(changed, op=dist.ReduceOp.MAX, group=group)
    if int(changed.item()):
        mine = torch.frombuffer(bytearray(handle), dtype=torch.uint8).to(device)
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        handles = [bytes(p.cpu().numpy().tobytes()) for p in parts]
        # one rank maps its peers at a time: four processes opening each other's handles at the same moment were seen to
        # block each other for good (each open is served by the exporting process); (re)attaching is rare, a few barriers
        # are not
        for turn in range(world):
            if turn == rank and ok:
                if not _ipc_probe_ok(shard, handles, rank):
                    ok, why = 0, "a probe process could not map a peer's slot array within its time limit"
                else:
                    try:
                        shard.attach_ipc(handles, rank)
                    except RuntimeError as e:          # either every rank is attached or none
                        ok, why = 0, str(e)
            dist.barrier(group=group)
        okt = torch.tensor([ok], dtype=torch.int64, device=device)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN, group=group)
        if not int(okt.item()):
            shard.detach()
            shard._attached_sig = None
            raise ShardAttachError("owner-sharded table: mapping the peers' slot arrays failed on some rank" + (": " + why if why else ""))
        shard._attached_sig = sig
    # 4. nobody reads a shard before every owner has finished writing its own
    _sync(device)
    dist.barrier(group=group)
    # every rank is attached to the present slot arrays (or still to the unchanged ones): arrays that were outgrown after
    # their handles had been given out can go now
    if hasattr(shard, "release_retired"):
        shard.release_retired()


def prefer_exchange(world, occurrences_per_rank, distinct_per_rank, link_gb_s=70.0, deduplicated=True):
    """Which way of getting a rank's counts to the key owners is expected to be faster ("which exchange"):
      local    count into a table of this rank's reads, ship its (hash, count) entries: 16 B per LOCAL distinct k-mer over
               the wire, ~33 ms of kernels per 10^9 occurrences (count + export by owner + the owner's add);
      exchange ship the region lists, deduplicated by the sender: 8 B x ~1.3 per LOCAL distinct k-mer, ~23 ms of kernels per
               10^9 occurrences, no local table.  Without the dedupe pass (tables too small to have second-level bits) the
               lists carry 8 B x ~1.2 per OCCURRENCE at ~20 ms of kernels.
    Every GPU talks to each of its world-1 peers over a link of its own (xGMI is point to point; ~70 GB/s per direction is
    assumed, nothing here has run on more than one GPU), so the bytes per link decide."""
    occ, dis = float(occurrences_per_rank), float(min(distinct_per_rank, occurrences_per_rank))
    per_link = 1.0 / max(world, 2)                       # each peer gets 1/world of what a rank produces
    t_local = 33e-3 * occ / 1e9 + 16.0 * dis * per_link / (link_gb_s * 1e9)
    if deduplicated:
        t_exchange = 23e-3 * occ / 1e9 + 8.0 * 1.3 * dis * per_link / (link_gb_s * 1e9)
    else:
        t_exchange = 20e-3 * occ / 1e9 + 9.6 * occ * per_link / (link_gb_s * 1e9)
    return t_exchange < t_local


XGMI_LINK_GB_S = 153.0       # one xGMI link, one direction (MI355X_MICROARCH.md: 7 links per GPU, point to point)


def dedupe_pays(world, occurrences_per_rank=None, link_gb_s=XGMI_LINK_GB_S, setting=None):
    """Should the sender deduplicate its region lists (list_dedupe_kernel) before the all_to_all?  The pass costs kernel time on
    every sender and saves bytes on the wire; with a link per peer the saving shrinks as the ranks grow.  From the role-play of
    round 4's kernels on the bench workload (profiles/round4/exchange_kernel_stats.csv; 1.07 G records per rank):

        ranks   dedupe pass   records left   raw wire / rank       wire saved          net
          2       5.3 ms        1 / 3.7       4.3 GB on 1 link     (4.3-1.4)/153 = 19 ms   +14 ms   -> dedupe
          4       5.4 ms        1 / 3.6       6.4 GB on 3 links    (6.4-2.1)/459 =  9 ms   + 4 ms   -> dedupe
          8       6.0 ms        1 / 2.4       7.5 GB on 7 links    (7.5-3.7)/1071 = 3.5 ms  - 2.5 ms -> ship the lists as they are

    (a rank's piece repeats fewer of its k-mers the more ranks share the reads, and more links carry what is left).  So: dedupe for
    2..4 ranks, not beyond -- until the pass runs hidden under the all_to_all of the round before, which is when the bytes alone
    would decide.  JASPER_AMD_EXCHANGE_DEDUPE=0 / 1 (`setting`) overrides.  Nothing here has run over xGMI."""
    if setting is None:
        setting = os.environ.get("JASPER_AMD_EXCHANGE_DEDUPE", "auto")
    if str(setting).lower() in ("0", "no", "false"):
        return False
    if str(setting).lower() in ("1", "yes", "true"):
        return True
    return world <= 4


def prefer_replicated(world, distinct_total, bases_per_rank, scans, free_bytes, polish_calls=1, link_gb_s=XGMI_LINK_GB_S):
    """After the counts have reached their key owners: keep the table sharded by owner (every lookup of the polishing kernels
    reads the owner's HBM, 7/8 of them over xGMI at N = 8), or give every GPU a copy of the whole table (merge: all-gather of the
    owners' entries + one import sweep; lookups local)?  The stated rule:

      it must FIT:   2 x 16 B x 2^ceil(log2(2 x distinct)) (the table at load <= 1/2, and the entries in flight) <= 80 % of the free HBM
      it must PAY:   t_gather + t_import < polish_calls x t_remote_penalty, with
          t_gather         = 16 B x distinct x (N-1)/N / ((N-1) links x link rate)        every rank receives the other owners' entries
          t_import         = distinct / 40e9                                              one region-ordered sweep (the rate of region_insert)
          t_remote_penalty = bases_per_rank x (N-1)/N x 1.2 x 64 B / ((N-1) links x link rate)      the dense scan of pass 0 probes every
                             window once (the later passes and the walks add ~0.07 lookups per base and scan: the 1.2), a 64-byte sector each
                           + scans x 2900 x (N-1)/N x 1.2 us        a pass is as long as its slowest segment's chain of dependent lookups
                             (2 900 in pass 0 of the bench workload, 0.24 us each locally; ~1.2 us more per remote round trip assumed)

    With BASELINE's shapes (distinct ~ 3..10 x assembly bases, one polish call of P + 1 scans per counted table) the gather costs more
    than all the remote lookups of the run -- the table stays sharded; a table that is polished many times over (a resident
    service, `polish_calls`) flips the rule.  configs[3] / [4] at 8 GPUs do not fit replicated at all.  Unmeasured on hardware."""
    n = max(int(world), 2)
    slots = 1
    while slots < 2 * max(int(distinct_total), 1):
        slots *= 2
    if 2 * 16 * slots > 0.8 * float(free_bytes):
        return False
    links = (n - 1) * link_gb_s * 1e9
    t_gather = 16.0 * distinct_total * (n - 1) / n / links
    t_import = distinct_total / 40e9
    t_penalty = bases_per_rank * (n - 1) / n * 1.2 * 64.0 / links + scans * 2900 * (n - 1) / n * 1.2e-6
    return t_gather + t_import < polish_calls * t_penalty


class _ResidentBases:
    """one buffer of bases in HBM, cut into pieces of at most `piece` bases"""

    def __init__(self, d_bases, n_bases, piece):
        self.ptr, self.n, self.piece, self.pos = int(d_bases), int(n_bases), int(piece), 0

    def next(self):
        if self.pos >= self.n:
            return None
        pos, end = self.pos, min(self.pos + self.piece, self.n)
        self.pos = end
        return self.ptr, self.n, pos, end, end < self.n

    def scanned(self):
        pass


class _FeedBases:
    """the batches of a read feed (KmerTable.feed_start): every batch is a buffer of its own, cut into pieces of at most
    `piece` bases; the reader goes on to parse the next batch as soon as the last piece of this one has been scanned"""

    def __init__(self, feeder, piece):
        self.feeder, self.piece, self.open = feeder, int(piece), True
        self.ptr = self.n = self.pos = 0

    def next(self):
        if not self.open:
            return None
        if self.pos >= self.n:
            self.ptr, self.n = self.feeder.feed_next()
            self.pos = 0
            if self.n == 0:
                self.open = False
                return None
        pos, end = self.pos, min(self.pos + self.piece, self.n)
        self.pos = end
        return self.ptr, self.n, pos, end, True

    def scanned(self):
        if self.pos >= self.n:
            self.feeder.feed_release()


def count_sharded(shard, d_bases, n_bases, device, group=None, piece_limit=None, clear=False, feeder=None, dedupe=None):
    """Count this rank's reads into the OWNER-SHARDED table without a table per GPU: every rank turns its reads into region
    lists grouped by the owner of the key (the two partition passes of the atomic-free counting path), ONE all_to_all per
    round moves every list to its owner (8 bytes per k-mer occurrence plus the slack of the lists), and the owner inserts what
    arrived straight into `shard` (include/jasper_hip.h: jasper_count_exchange_*).  Compared with counting into a local table
    and shard_tables(): no local insert, no export pass, no add pass, and no memory for a local table.

    The reads: text bases in HBM (d_bases, n_bases -- as for KmerTable.count_bases_device; cut into rounds of piece_limit
    bases), or `feeder` = a KmerTable on which feed_start(ranges) has been called: every batch the file reader delivers is a
    round, and the reader parses the next batch while the last one is exchanged.

    Collective.  `shard` should be sized for the keys it will own (min_slots / reserve) -- it grows if it must.  Counts are
    ADDED to what the shard holds; clear=True (rank-uniform: checked) empties it first (at a point where no peer can still be reading it).  Returns
    None (collectively, nothing consumed) when the table / k has no exchange geometry -- count into a local table and call
    shard_tables() then -- else a dict with the rounds and bytes moved.  Afterwards lookups through `shard` read the owner's HBM
    (own or peer's), as after shard_tables()."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        raise RuntimeError("count_sharded needs an initialised process group of more than one rank")
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    was_empty = clear or shard.info()["distinct"] == 0
    agree = torch.tensor([shard.info()["slots"], 0 if was_empty else 1, 1 if clear else 0, 0 if clear else 1], dtype=torch.int64, device=device)
    dist.all_reduce(agree, op=dist.ReduceOp.MAX, group=group)
    slots, any_filled, some_clear, some_keep = (int(v) for v in agree.tolist())
    if some_clear and some_keep:
        # `clear` adds a reduction below that only the clearing ranks would enter: ranks that disagree on it would hang there.  Every
        # rank sees the same two maxima, so every rank leaves here together.
        raise ValueError("count_sharded: clear must be the same on every rank")
    # (every rank is here: none of them still reads the shards of the last step through its peer mappings)
    # Every step below that can fail on ONE rank (a device allocation of reserve(), a library call) reports into the next
    # reduction instead of raising: a rank that left through an exception while its peers wait in a collective hangs the job.
    ok, why, plan = 1, "", None
    try:
        shard.reserve(slots)
        plan = shard.exchange_plan(1 << 26, world)      # (whether there is a geometry does not depend on the piece size)
    except RuntimeError as e:
        ok, why = 0, str(e)
    okt = torch.tensor([1 if plan is not None else 0, ok], dtype=torch.int64, device=device)
    dist.all_reduce(okt, op=dist.ReduceOp.MIN, group=group)
    if not int(okt[1].item()):
        raise CollectiveCountError("count_sharded: preparing the shards failed on some rank" + (": " + why if why else ""))
    if not int(okt[0].item()):
        return None
    # only now, when EVERY rank has a geometry and the call will go through, is a shard emptied: a caller that keeps its shard
    # after a `None` (some peer has no plan) or a CollectiveCountError must find its counts as they were on every rank
    if clear:
        ok, why = 1, ""
        try:
            shard.clear()
        except RuntimeError as e:
            ok, why = 0, str(e)
        okc = torch.tensor([ok], dtype=torch.int64, device=device)
        dist.all_reduce(okc, op=dist.ReduceOp.MIN, group=group)
        if not int(okc.item()):
            raise CollectiveCountError("count_sharded: emptying the shards failed on some rank" + (": " + why if why else ""))
    # a round holds its send and receive lists (~10 bytes per base each) next to the shard: at most 2^31 bases, fewer when the
    # memory is short (one value for all ranks: the buffers are sized by the longest piece of the round).  The gloo rehearsal
    # gathers every rank's send buffer on every rank, and its ranks may share one GPU.
    if dedupe is None:
        dedupe = dedupe_pays(world)
    # THE ROUNDS ARE A PIPELINE OF THREE STAGES (round 5).  A round's work is: SENDER -- this rank's piece through part1 and the split by
    # owner (+ dedupe), kernels on the library's stream; WIRE -- the all_to_all of the lists; OWNER -- region_insert of what arrived.
    # Run one after the other a rank's GPU idles during the wire and its links idle during the kernels (~24 ms per rank at
    # N = 8 where the kernels are ~15).  So the all_to_all of round r is started without waiting for it (async_op), the sender stage of
    # round r + 1 runs while it is under way, and only then is it waited for and inserted -- behind the START of round r + 1's own
    # all_to_all, which is then under way during that insert and during the sender stage of round r + 2.  A rank's kernels stay on one
    # stream (they would only share the chip), what overlaps is the fabric with the kernels.
    # The small agreements between the stages (sizes, record counts, "did every rank get through") must not queue behind a list
    # exchange on the same communicator: they go through a CONTROL group of their own on host tensors (gloo).
    pipeline = os.environ.get("JASPER_AMD_EXCHANGE_PIPELINE", "1") not in ("0", "no", "false")
    ctrl = _control_group(group) if pipeline else group
    cdev = torch.device("cpu") if (pipeline and ctrl is not group) or dist.get_backend(group) != "nccl" else device

    def agree(values, op):
        t = torch.tensor(list(values), dtype=torch.int64, device=cdev)
        dist.all_reduce(t, op=op, group=ctrl)
        return [int(v) for v in t.tolist()]

    def gather_rows(values):
        t = torch.tensor(list(values), dtype=torch.int64, device=cdev)
        parts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(parts, t, group=ctrl)
        return [q.tolist() for q in parts]

    piece = int(piece_limit or os.environ.get("JASPER_AMD_EXCHANGE_PIECE", 1 << 31))
    if device.type == "cuda" and not piece_limit:
        torch.cuda.empty_cache()
        # a round holds its send and receive lists (~10 bytes per base each) next to the library's level-1 lists; with the pipeline the
        # lists of up to three rounds exist at once (being made / on the wire / being inserted)
        per_base = (72 if pipeline else 48) if dist.get_backend(group) == "nccl" else (72 if pipeline else 48) + 48 * world
        piece = max(8 << 20, min(piece, torch.cuda.mem_get_info(device)[0] // per_base))
    longest = int(n_bases) if feeder is None else 0
    piece, longest = agree([piece, -longest], dist.ReduceOp.MIN)
    longest = -longest
    if pipeline and feeder is None and not piece_limit and longest <= piece and longest >= (1 << 30):
        # bases resident in HBM that would be ONE round: three rounds, so that two thirds of the lists travel under kernels
        # (a round costs ~1 ms of launches and agreements of its own: not worth it for small inputs)
        piece = (longest + 2) // 3 + 64
    source = _FeedBases(feeder, piece) if feeder is not None else _ResidentBases(d_bases, n_bases, piece)
    wire = n_deferred = rounds = 0
    records_max = 0
    plan = None
    last = {}

    def sender_stage(index):
        """this rank's next piece -> send lists; None (on every rank together) when no rank has a piece left"""
        nonlocal records_max, plan
        ok, why = 1, ""
        mine = None
        try:
            mine = source.next()
        except RuntimeError as e:                       # (a reader / parser error of the feed)
            ok, why = 0, str(e)
        length = (mine[3] - mine[2]) if mine else 0
        more = 1 if (mine and mine[4]) else 0
        piece_max, slots, any_more, bad = agree([length, shard.info()["slots"], more, 1 - ok], dist.ReduceOp.MAX)
        if bad:
            raise CollectiveCountError("count_sharded: reading the reads failed on some rank" + (": " + why if why else ""))
        if piece_max == 0:
            return None
        plan = None
        try:
            shard.reserve(slots)                        # a shard that grew in the last round changes the geometry for all
            plan = shard.exchange_plan(piece_max, world)
        except RuntimeError as e:
            ok, why = 0, str(e)
        has_plan, all_ok = agree([1 if plan is not None else 0, ok], dist.ReduceOp.MIN)
        if not all_ok:
            raise CollectiveCountError("count_sharded: growing the shards failed on some rank" + (": " + why if why else ""))
        if not has_plan:
            raise CollectiveCountError("count_sharded: the shards outgrew the exchange geometry between rounds")
        dcap = plan["deferred_cap"]
        found = 0
        deferred = None
        try:                                            # first pass: my reads -> level-1 lists inside the library; how many records?
            # (libjasper_hip works on its own stream: torch memory must be idle before it is handed over -- empty(), never zeros())
            deferred = torch.empty(8 + 3 * dcap, dtype=torch.int64, device=device)
            _sync(device)
            if mine:
                found = shard.exchange_scan(mine[0], mine[1], mine[2], mine[3], piece_max, world, deferred.data_ptr(), dcap)
                source.scanned()
            else:
                found = shard.exchange_scan(0, 0, 0, 0, piece_max, world, deferred.data_ptr(), dcap)
        except RuntimeError as e:
            ok, why = 0, str(e)
        found_max, failed = agree([found, 1 - ok], dist.ReduceOp.MAX)
        if failed:
            raise CollectiveCountError("count_sharded: the first partition pass failed on some rank" + (": " + why if why else ""))
        records_max = max(found_max, 1)                 # the send lists are sized from the records that are really there
        send = send_cnt = None
        nrec = ncnt = 0
        try:                                            # second pass: level-1 lists -> region lists grouped by owner
            plan = shard.exchange_plan(piece_max, world, records_max)
            nrec, ncnt = plan["records_per_owner"], plan["counts_per_owner"]
            send = torch.empty((world, nrec), dtype=torch.int64, device=device)
            send_cnt = torch.empty((world, ncnt), dtype=torch.int32, device=device)
            _sync(device)
            shard.exchange_partition(piece_max, records_max, world, send.data_ptr(), send_cnt.data_ptr(), deferred.data_ptr(), dcap)
            shard.sync()
        except (RuntimeError, TypeError) as e:          # (TypeError: no plan for these sizes -- exchange_plan returned None)
            ok, why = 0, str(e)
        ndef = int(deferred[0].item()) if ok else 0
        if ndef > dcap:
            ok, why = 0, "too many records found no room in their lists (%d): pass a larger size hint" % ndef
        # third pass (optional): a read shard repeats its k-mers, and all copies of one are in the same list -- the lists are
        # deduplicated in place (one record per distinct key, its occurrences in bits the list implies) and only the filled part
        # of every list, as long as the fullest list of any rank, travels
        fill, cbits = 0, 0
        if ok and dedupe and plan["p2"] >= 1:
            try:
                dd = shard.exchange_dedupe(piece_max, records_max, world, send.data_ptr(), send_cnt.data_ptr())
                if dd is not None:
                    fill, cbits = dd
            except RuntimeError as e:
                ok, why = 0, str(e)
        parts = gather_rows([ndef, 1 - ok, fill])
        if any(q[1] for q in parts):
            raise CollectiveCountError("count_sharded: partitioning failed on some rank" + (": " + why if why else ""))
        slice_cap = 0
        if cbits:
            slice_cap = max(max(q[2] for q in parts), 1)
            send = send.view(world * ncnt, plan["slice_cap"])[:, :slice_cap].contiguous().view(world, ncnt * slice_cap)
            nrec = ncnt * slice_cap
        _sync(device)                                   # (the lists are complete and packed: they may travel)
        return dict(send=send, send_cnt=send_cnt, deferred=deferred, parts=parts, piece_max=piece_max, records_max=records_max, slice_cap=slice_cap, cbits=cbits,
                    nrec=nrec, ncnt=ncnt, slots=shard.info()["slots"], whole=(index == 0 and not any_more and not any_filled), plan=plan)

    def start_wire(R):
        """the lists, their fill counts and (rare) the deferred entries of all ranks on their way; nothing is waited for"""
        nonlocal wire, n_deferred
        R["recv"], w1 = _all_to_all_rows_async(R["send"], group)
        R["recv_cnt"], w2 = _all_to_all_rows_async(R["send_cnt"], group)
        R["works"] = [w1, w2]
        wire += (world - 1) * (R["nrec"] * 8 + R["ncnt"] * 4)
        R["d_all"], R["n_all"] = None, 0
        mx = max(q[0] for q in R["parts"])
        if mx:                                          # rare: the deferred entries of all ranks go to everybody, owners pick theirs
            pad = R["deferred"][8:8 + 3 * mx].contiguous()
            allp = [torch.empty_like(pad) for _ in range(world)]
            R["works"].append((lambda w: (lambda: w.wait()))(dist.all_gather(allp, pad, group=group, async_op=True)))
            R["allp"], R["n_all"] = allp, sum(q[0] for q in R["parts"])
            n_deferred += R["n_all"]

    def owner_stage(R):
        """what arrived for this owner into its shard"""
        ok, why = 1, ""
        for w in R["works"]:
            w()
        _sync(device)
        recv, recv_cnt = R["recv"](), R["recv_cnt"]()
        d_all = None
        if R["n_all"]:
            d_all = torch.cat([R["allp"][src][:3 * R["parts"][src][0]] for src in range(world)]).contiguous()
        _sync(device)
        R["send"] = R["send_cnt"] = None
        try:
            if shard.info()["slots"] != R["slots"]:     # (a shard that grew while these lists were on their way: they are lists of the old geometry)
                raise RuntimeError("a shard grew between the split of a round and its insert")
            shard.exchange_insert(recv.data_ptr(), recv_cnt.data_ptr(), R["piece_max"], R["records_max"], world, rank, d_all.data_ptr() if R["n_all"] else 0, R["n_all"],
                                  whole_input=R["whole"], slice_cap=R["slice_cap"], count_bits=R["cbits"])
            shard.sync()
        except RuntimeError as e:
            ok, why = 0, str(e)
        if not agree([ok], dist.ReduceOp.MIN)[0]:
            raise CollectiveCountError("count_sharded: inserting the received lists failed on some rank" + (": " + why if why else ""))
        last.update(plan=R["plan"], slice_cap=R["slice_cap"], cbits=R["cbits"])

    in_flight = None
    while True:
        cur = sender_stage(rounds + (1 if in_flight is not None else 0))
        if cur is not None:
            start_wire(cur)
        if not pipeline and cur is not None:
            owner_stage(cur)
            rounds += 1
            continue
        if in_flight is not None:
            owner_stage(in_flight)
            rounds += 1
        in_flight = cur
        if cur is None:
            break
    plan = last.get("plan", plan)
    slice_cap, cbits = last.get("slice_cap", 0), last.get("cbits", 0)
    if device.type == "cuda" and rounds > 1:
        torch.cuda.empty_cache()                        # (rounds of different sizes leave cached blocks behind: back to the driver)
    _attach_shards(shard, device, group)
    return dict(rounds=rounds, wire_bytes=wire, deferred=n_deferred, plan=plan, records_max=records_max, deduplicated=bool(cbits), slice_cap_sent=slice_cap or plan["slice_cap"])


def write_jf_sharded(shard, path, cmdline, device, group=None):
    """`mer_counts$K.jf` from an owner-sharded table (src/jasper.sh:177 `tee $JF_DB`): a binary/sorted file is ordered by
    the key rotated right by log2(size) bits, so the value range of the key's low bits cut into `world` equal parts gives
    `world` consecutive pieces of the file.  Every rank groups its shard's entries by piece (one table pass), ONE all_to_all
    brings each piece together on one GPU, which sorts it and writes `<path>.piece<r>`; rank 0 writes the header and strings
    the pieces together.  Collective; the file is complete when it returns."""
    import shutil
    import torch
    import torch.distributed as dist
    from .table import KmerTable
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    k = shard.k
    size_log2 = min(2 * k, 60, (shard.info()["slots"] * world - 1).bit_length())
    est = torch.tensor([int(shard.info()["distinct"] / world * 1.1) + (1 << 16)], dtype=torch.int64, device=device)
    dist.all_reduce(est, op=dist.ReduceOp.MAX, group=group)
    mx = int(est.item())
    send = torch.empty((world, mx, 2), dtype=torch.int64, device=device)
    _sync(device)
    ct = torch.tensor(shard.export_file_ranges(send.data_ptr(), mx, world, size_log2), dtype=torch.int64, device=device)
    allc = [torch.zeros_like(ct) for _ in range(world)]
    dist.all_gather(allc, ct, group=group)
    allc = torch.stack(allc).cpu()                     # allc[src][piece]
    if int(allc.max().item()) > mx:                    # (key ends are not as even as hashes: exact sizes when the guess was low)
        mx = int(allc.max().item())
        del send
        send = torch.empty((world, mx, 2), dtype=torch.int64, device=device)
        _sync(device)
        shard.export_file_ranges(send.data_ptr(), mx, world, size_log2)
    _sync(device)
    recv = _all_to_all_rows(send, group)
    _sync(device)
    del send
    incoming = int(allc[:, rank].sum().item())
    piece = KmerTable(k, min_slots=max(1 << 16, 2 * incoming), device=shard.device)
    srcs = [src for src in range(world) if int(allc[src][rank])]
    if srcs:
        piece.import_packed_multi([recv[src].data_ptr() for src in srcs], [int(allc[src][rank]) for src in srcs])
    del recv
    piece.write_jf_piece("%s.piece%d" % (path, rank), [], size_log2, 1)
    if rank == 0:
        piece.write_jf_piece(path + ".tmp", list(cmdline), size_log2, 2)
    piece.close()
    _sync(device)
    dist.barrier(group=group)
    if rank == 0:
        with open(path + ".tmp", "ab") as out:
            for r in range(world):
                with open("%s.piece%d" % (path, r), "rb") as f:
                    shutil.copyfileobj(f, out, 16 << 20)
                os.remove("%s.piece%d" % (path, r))
        os.replace(path + ".tmp", path)
    dist.barrier(group=group)


def histogram_sharded(shard, device, group=None):
    """histogram of an owner-sharded table: every owner bins its own keys, the 10002 bins are summed over ranks"""
    import torch
    import torch.distributed as dist
    h = torch.tensor(shard.histogram(), dtype=torch.int64, device=device)
    dist.all_reduce(h, group=group)
    return [int(x) for x in h.tolist()]


def histogram_merged(table, device, group=None):
    """histogram of a table that merge_tables has made identical on all ranks: every rank bins the key range it owns
    (1/world of a table pass), the 10002 bins are summed over ranks"""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return table.histogram()
    world = dist.get_world_size(group)
    h = torch.tensor(table.histogram_part(dist.get_rank(group), world), dtype=torch.int64, device=device)
    dist.all_reduce(h, group=group)
    return [int(x) for x in h.tolist()]


def all_reduce_ints(values, device=None, op="sum"):
    """sum (or "min" / "max") a short list of python ints over ranks (QV counters, k-mer totals, agreed decisions)"""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return list(values)
    t = torch.tensor(list(values), dtype=torch.int64, device=device)
    dist.all_reduce(t, op={"sum": dist.ReduceOp.SUM, "min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX}[op])
    return [int(x) for x in t.tolist()]
