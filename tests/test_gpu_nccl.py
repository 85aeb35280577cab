"""GPU, one rank PER DEVICE, backend nccl (= RCCL over xGMI): the N>1 path on real multi-GPU hardware.

Skipped only when the box has fewer than two GPUs (the 1-GPU test box rehearses the same code with both ranks on its one
GPU and gloo as the transport: test_gpu_shard.py, test_gpu_dist2.py, test_gpu_cli_e2e.py).  What runs here that cannot
run there: dist._all_to_all_rows' all_to_all_single branch, RCCL all_reduce / all_gather on device tensors, and lookups
that read a PEER GPU's slot array through an IPC mapping over xGMI.
Role replaced: the reference's `xargs -P` fan-out (src/jasper.sh:212) and `jellyfish merge`
(JF::jellyfish/merge_files.cc:44-176)."""
import gzip
import json
import os
import re
import shutil
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))
E2E = os.path.join(HERE, "golden", "e2e")


def _ngpu():
    import torch
    return torch.cuda.device_count()      # (counting devices does not initialise HIP)


needs2 = pytest.mark.skipif(_ngpu() < 2, reason="needs two GPUs (one rank per device, backend nccl)")


def _exchange_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from jasper_amd import dist as jd
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        # send[dst] = rows tagged (src, dst, i): after the exchange recv[src] must carry (src, me, i)
        send = torch.empty((world, 5, 2), dtype=torch.int64, device=dev)
        for dst in range(world):
            for i in range(5):
                send[dst, i, 0] = rank * 1000 + dst
                send[dst, i, 1] = i
        recv = jd._all_to_all_rows(send)
        ok = all(int(recv[src, i, 0]) == src * 1000 + rank and int(recv[src, i, 1]) == i for src in range(world) for i in range(5))
        tot = jd.all_reduce_ints([rank + 1, 10], device=dev)
        q.put((rank, ok, list(tot), dist.get_backend()))
    finally:
        dist.destroy_process_group()


@needs2
def test_entry_exchange_over_rccl(hip):
    import torch.multiprocessing as mp
    from test_gpu_shard import _free_port
    world = min(_ngpu(), 4)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_exchange_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=300) for _ in ps], key=lambda x: x[0])
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r, ok, tot, backend in res:
        assert ok and backend == "nccl"
        assert tot == [world * (world + 1) // 2, 10 * world]


@needs2
def test_shard_tables_and_polish_over_rccl_equal_single_gpu(hip):
    """dist.shard_tables (one all_to_all_single over RCCL, owners' slot arrays IPC-mapped between DIFFERENT GPUs) + polish
    through the sharded view == one table on one GPU: histogram, disjoint cover of the keys, polished text, QV counters"""
    import torch.multiprocessing as mp
    import test_gpu_shard as S
    from jasper_amd import KmerTable, synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = S._free_port()
    ps = [ctx.Process(target=S._worker, args=(r, 2, port, q, "nccl", False)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=300) for _ in ps], key=lambda x: x[0])
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    k = 37
    rng = np.random.default_rng(43)
    genome = synth.make_genome(rng, 150_000)
    reads = synth.make_reads_stream(rng, genome, 30, 150, 0.003)
    asm = synth.make_assembly(rng, genome, err=1e-3, n_every=10**9).tobytes().decode()
    t = KmerTable(k, min_slots=1 << 21, device=0)
    t.count_bases(reads.tobytes())
    h = t.histogram()
    recs = synth.chunk_records("c", len(asm), 20_000)
    full = t.polish_batch([asm[a:b] for _, a, b in recs], 3, 2)
    for step in range(2):
        r0, r1 = res[0][1][step], res[1][1][step]
        assert r0[0] == h and r1[0] == h
        assert r0[1] + r1[1] == t.info()["distinct"]
        assert r0[6] == r1[6]
        got = [None] * len(recs)
        for r in (r0, r1):
            for i, s in zip(r[3], r[4]):
                got[i] = s
        assert got == full.seqs
        assert tuple(r0[5]) == tuple(r1[5]) == full.qv
    t.close()


@needs2
def test_replicated_merge_over_rccl_equals_single_gpu(hip):
    """dist.merge_tables (reduce-scatter by key range + all_gather, both over RCCL) leaves the full counts on every GPU"""
    import torch.multiprocessing as mp
    import test_gpu_dist2 as D
    from jasper_amd import KmerTable, synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = D._free_port()
    ps = [ctx.Process(target=D._worker, args=(r, 2, port, q, "nccl", False)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=300) for _ in ps], key=lambda x: x[0])
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    rng = np.random.default_rng(42)
    genome = synth.make_genome(rng, 150_000)
    reads = synth.make_reads_stream(rng, genome, 30, 150, 0.003)
    t = KmerTable(37, min_slots=1 << 21, device=0)
    t.count_bases(reads.tobytes())
    h = t.histogram()
    assert res[0][1] == h and res[1][1] == h
    assert res[0][2] == res[1][2] == t.info()["distinct"]
    t.close()


@needs2
@pytest.mark.parametrize("count", ["local", "exchange"])
def test_cli_two_gpus_matches_jasper_sh(hip, tmp_path, count):
    """`python -m jasper_amd.cli --gpus 2 ...`: the driver starts its own two ranks (one per GPU, RCCL) -- same artefacts
    as the real jasper.sh run kept under tests/golden/e2e, with the counts travelling as table entries or as region lists"""
    from test_gpu_cli_e2e import fasta_records
    meta = json.load(open(os.path.join(E2E, "meta.json")))
    for fn in ("r1.fq", "r2.fq"):
        with open(tmp_path / fn, "wb") as f:
            f.write(gzip.open(os.path.join(E2E, fn + ".gz")).read())
    shutil.copy(os.path.join(E2E, "asm.fa"), tmp_path)
    env = dict(os.environ, PYTHONPATH=ROOT, JASPER_AMD_COUNT=count, JASPER_AMD_TIMING="1")
    for v in ("JASPER_AMD_DIST_BACKEND", "JASPER_AMD_ONE_GPU"):
        env.pop(v, None)
    p = subprocess.run([sys.executable, "-m", "jasper_amd.cli", "--gpus", "2", "-r", "r1.fq r2.fq", "-a", "asm.fa", "-k", str(meta["k"]),
                        "-t", str(meta["threads"]), "-p", str(meta["passes"])], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert ("region lists -> owners' shards" in p.stderr) == (count == "exchange"), p.stderr
    assert open(tmp_path / "threshold.txt").read() == open(os.path.join(E2E, "threshold.txt")).read()
    assert open(tmp_path / "jfhisto25.csv").read() == open(os.path.join(E2E, "jfhisto25.csv")).read()
    assert fasta_records(tmp_path / "asm.fa.polished.fasta") == fasta_records(os.path.join(E2E, "asm.fa.polished.fasta"))
    assert open(tmp_path / "asm.fa.fixes.csv", newline="").read() == open(os.path.join(E2E, "asm.fa.fixes.csv"), newline="").read()
    mine = [re.sub(r"^\[[^\]]*\]", "[DATE]", ln) for ln in p.stdout.splitlines() if re.match(r"^\[\w{3} \w{3} +\d", ln)]
    strip_q = lambda ls: [re.sub(r"Q value = .*", "Q value =", ln) for ln in ls]
    assert strip_q(mine) == strip_q(meta["stdout"])


@needs2
@pytest.mark.parametrize("count", ["auto", "exchange"])
def test_bench_two_gpus_self_launch(hip, count):
    """`python bench.py --gpus 2` starts its own ranks and prints ONE line that says which world it saw; with --count exchange
    the region lists travel over RCCL instead of the entries of per-GPU tables"""
    env = dict(os.environ, PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--genome-mb", "4", "--count", count],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_world_size"] == 2 and out["backend"] == "nccl"
    assert out["rccl_selftest"] and out["rccl_selftest"]["ok"] and out["rccl_selftest"]["world"] == 2      # first contact made in throw-away processes
    assert "table" in out["config"] and out["value"] > 0
    if count == "exchange":
        assert out["count_exchange"] is not None
