"""GPU, 2 processes sharing the one GPU of the test box: the N>1 data path end to end (read shard -> count -> export ->
exchange -> import -> identical merged tables -> chunk shard -> polish) with gloo as the transport, because RCCL refuses
two ranks on one device.  Everything except the transport is what bench.py / the 8-GPU run executes."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, backend="gloo", one_gpu=True):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from jasper_amd import KmerTable, synth, dist as jd
    di = 0 if one_gpu else rank                          # one rank per device when the box has them (backend nccl = RCCL)
    if backend == "nccl":
        torch.cuda.set_device(di)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", di))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        k = 37
        rng = np.random.default_rng(42)                     # same workload on every rank
        genome = synth.make_genome(rng, 150_000)
        reads = synth.make_reads_stream(rng, genome, 30, 150, 0.003)
        asm = synth.make_assembly(rng, genome, err=1e-3, n_every=10**9).tobytes().decode()
        nrec = reads.size // 151
        lo, hi = jd.shard_range(nrec, rank, world)          # read shard of this rank
        mine = reads[lo * 151:hi * 151].tobytes()
        t = KmerTable(k, min_slots=1 << 21, device=di)
        t.count_bases(mine)
        dev = torch.device("cuda", di)
        merged = jd.merge_tables(t, dev)
        h = t.histogram()
        assert jd.histogram_merged(t, dev) == h             # owner ranges binned per rank + all_reduce == full scan
        info = t.info()
        # chunk shard
        bs = 20_000
        recs = synth.chunk_records("c", len(asm), bs)
        owner = jd.assign_chunks([b - a for _, a, b in recs], world)
        my = [i for i, o in enumerate(owner) if o == rank]
        res = t.polish_batch([asm[recs[i][1]:recs[i][2]] for i in my], 3, 2)
        qv = jd.all_reduce_ints(list(res.qv), device=dev)
        q.put((rank, h, info["distinct"], merged, my, res.seqs, qv))
        t.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu(hip):
    import torch.multiprocessing as mp
    from jasper_amd import KmerTable, synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=300) for _ in ps], key=lambda x: x[0])
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    # single-table reference on the same workload
    k = 37
    rng = np.random.default_rng(42)
    genome = synth.make_genome(rng, 150_000)
    reads = synth.make_reads_stream(rng, genome, 30, 150, 0.003)
    asm = synth.make_assembly(rng, genome, err=1e-3, n_every=10**9).tobytes().decode()
    t = KmerTable(k, min_slots=1 << 21, device=0)
    t.count_bases(reads.tobytes())
    h = t.histogram()
    assert res[0][1] == h and res[1][1] == h                      # both ranks hold the full counts after the merge
    assert res[0][2] == res[1][2] == t.info()["distinct"]
    assert res[0][3] > 0 and res[1][3] > 0
    recs = synth.chunk_records("c", len(asm), 20_000)
    full = t.polish_batch([asm[a:b] for _, a, b in recs], 3, 2)
    got = [None] * len(recs)
    for r in res:
        for i, s in zip(r[4], r[5]):
            got[i] = s
    assert got == full.seqs                                       # chunk shards together == unsharded run
    assert tuple(res[0][6]) == tuple(res[1][6]) == full.qv        # QV counters all-reduced
    t.close()


def test_transport_selftest_in_throwaway_processes(hip):
    """dist.transport_selftest: what `bench.py --gpus N` does before it commits to RCCL -- the process group, one all_to_all of
    64 MB checked word by word, all_reduce / all_gather / barrier -- in a child job under a time limit.  On the one GPU of the
    test box the transport is gloo (RCCL refuses two ranks on one device); with two GPUs, test_gpu_nccl.py runs bench.py over
    RCCL with the self-test in front.  A job that does not answer in time is killed and reported, never waited for."""
    sys.path.insert(0, ROOT)
    from jasper_amd import dist as jd
    res = jd.transport_selftest(2, backend="gloo", one_gpu=True, seconds=240, mb=16)
    assert res["ok"], res
    assert res["world"] == 2 and res["ms"]["all_to_all_single"] > 0 and res["bytes_all_to_all"] >= (15 << 20)
    # ... and the pattern of the pipelined exchange: an all_to_all that is not waited for, agreements on the control group beside it
    assert res["pipeline_ok"] and res["ms"]["async_all_to_all_with_agreements"] > 0, res
    res = jd.transport_selftest(2, backend="gloo", one_gpu=True, seconds=1, mb=16)      # (two interpreters do not even start in 1 s)
    assert not res["ok"] and "killed" in res["error"] and "did not go away" not in res["error"], res
    # nothing of the killed job is left behind (the launcher starts every rank in a session of its own)
    left = []
    for pid in os.listdir("/proc"):
        if pid.isdigit():
            try:
                args = open("/proc/%s/cmdline" % pid, "rb").read().split(b"\0")
            except OSError:
                continue
            if b"jasper_amd._selftest" in args:
                left.append((pid, args))
    assert not left, left


def test_bench_three_ranks_rehearsed_on_one_gpu(hip):
    """`bench.py --gpus 3 --backend gloo --one-gpu` at small size: the N-rank driver path (read shards, list exchange or entries,
    owner-sharded table over hipIpc, chunk shards, max-over-ranks timing, ONE json line) with three processes on the one GPU of the
    test box (it lets six processes use its card at once: the ranks, this test runner, one IPC probe at a time); the 8-rank protocol runs on the
    CPU in test_dist_gloo.py"""
    import json
    import subprocess
    env = dict(os.environ, PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "1", "--warmup", "1", "--genome-mb", "2", "--backend", "gloo", "--one-gpu"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 3 and out["rccl_world_size"] == 3 and out["backend"] == "gloo" and out["value"] > 0
    assert out["rccl_selftest"] is None                        # (the self-test is RCCL's: not run for the gloo rehearsal)
