"""GPU: the owner-sharded table ("keep the table key-sharded and route lookups").  Owner o of n keeps only
the keys with owner_of(hash) == o; every lookup reads the owner's slot array.  Checked bit-exact against the whole table:
  * one process, n shard tables on the one GPU (attach_tables): export grouped by owner is a disjoint cover, lookups /
    histogram / polishing through the shards == through the whole table, for n = 2, 3, 8 and k on both hash paths;
  * two processes on the one GPU (dist.shard_tables with gloo as the transport, because RCCL refuses two ranks on one
    device): count shards -> one all_to_all -> owners add -> peers' slot arrays IPC-mapped -> polish == unsharded run.
"""
import os
import socket
import sys

import numpy as np
import pytest

from jasper_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def KT(hip):
    from jasper_amd import KmerTable
    return KmerTable


def workload(seed, G, k, cov=30, rl=150, err=0.003, asm_err=1e-3):
    rng = np.random.default_rng(seed)
    genome = synth.make_genome(rng, G)
    reads = synth.make_reads_stream(rng, genome, cov, rl, err)
    asm = synth.make_assembly(rng, genome, err=asm_err, n_every=max(G // 3, 1000), n_len=60)
    return genome, reads.tobytes(), asm.tobytes().decode()


def make_shards(KT, full, n, slots):
    """split `full` into n owner tables of `slots` slots each (what dist.shard_tables does, without the exchange)"""
    import torch
    from jasper_amd import _lib
    L = _lib.lib()
    distinct = full.info()["distinct"]
    cap = int(distinct / n * 1.2) + 4096
    buf = torch.empty((n, cap, 2), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    counts = full.export_owner(buf.data_ptr(), cap, n)
    assert sum(counts) == distinct and max(counts) <= cap
    # the grouping is owner_of of each entry's hash, and the groups are even
    host = buf.cpu().numpy()
    for o in range(n):
        seg = host[o, :counts[o]]
        for lo, hi in seg[:: max(1, counts[o] // 200)]:
            sh = max(0, 2 * full.k - 64)
            hi_bits = int(hi) & ((1 << sh) - 1) if sh else 0
            assert L.jasper_owner_of(int(lo) & (2**64 - 1), hi_bits, n) == o
        assert abs(counts[o] - distinct / n) < 0.05 * distinct / n + 2000
    shards = []
    for o in range(n):
        t = KT(full.k, min_slots=slots)
        if o % 2:
            t.import_packed(buf[o].data_ptr(), counts[o], 0)
        else:        # the one-sweep form, the list cut into three uneven pieces (one of them empty)
            c1, c2 = counts[o] // 5, counts[o] // 2
            t.import_packed_multi([buf[o].data_ptr(), buf[o][c1:].data_ptr(), buf[o][c2:].data_ptr(), buf[o].data_ptr()], [c1, c2 - c1, counts[o] - c2, 0])
        assert t.info()["slots"] == slots, "the test sizes the shards so that they do not grow"
        assert t.info()["distinct"] == counts[o]
        shards.append(t)
    return shards, counts


@pytest.mark.parametrize("k,n", [(37, 2), (37, 3), (37, 8), (25, 4), (32, 2)])
def test_lookups_histogram_polish_through_shards_equal_whole_table(KT, k, n):
    G = 200_000
    genome, reads, asm = workload(100 + k + n, G, k)
    full = KT(k, min_slots=1 << 21)
    full.count_bases(reads)
    shards, counts = make_shards(KT, full, n, 1 << 21)
    for o, t in enumerate(shards):
        t.attach_tables(shards, o)
    # lookups: windows of the assembly (present, absent, with N) through every shard's view
    rng = np.random.default_rng(5)
    pos = rng.integers(0, len(asm) - k, 20_000)
    qs = [asm[p:p + k] for p in pos]
    want = full.lookup(qs)
    assert sum(1 for c in want if c) > 10_000
    for t in shards:
        assert t.lookup(qs) == want
    # histogram: owners bin their own keys
    acc = [0] * 10002
    for t in shards:
        acc = [a + b for a, b in zip(acc, t.histogram())]
    assert acc == full.histogram()
    # polishing through a sharded view == through the whole table (text, every fix record, QV counters)
    recs = synth.chunk_records("c", len(asm), 30_000)
    chunks = [asm[a:b] for _, a, b in recs]
    ref = full.polish_batch(chunks, 3, 2)
    for t in (shards[0], shards[-1]):
        got = t.polish_batch(chunks, 3, 2)
        assert got.seqs == ref.seqs
        assert got.qv == ref.qv
        assert got.records == ref.records
    # detaching gives back the owner's own keys only
    shards[0].detach()
    own = shards[0].lookup(qs)
    assert all(c == w or c == 0 for c, w in zip(own, want)) and own != want
    for t in shards + [full]:
        t.close()


def test_attach_rejects_mismatched_geometry(KT):
    a, b = KT(37, min_slots=1 << 21), KT(37, min_slots=1 << 22)
    with pytest.raises(RuntimeError, match="same k and slot count"):
        a.attach_tables([a, b], 0)
    c = KT(25, min_slots=1 << 21)
    with pytest.raises(RuntimeError, match="same k and slot count"):
        a.attach_tables([a, c], 0)
    for t in (a, b, c):
        t.close()


def test_growing_an_attached_table_detaches_it(KT):
    k = 37
    _, reads, asm = workload(9, 60_000, k)
    a, b = KT(k, min_slots=1 << 16), KT(k, min_slots=1 << 16)
    a.attach_tables([a, b], 0)
    a.count_bases(reads)                 # far more keys than 2^16 slots: grows, which must drop the stale shard view
    qs = [asm[i:i + k] for i in range(0, 5000, 7)]
    whole = KT(k, min_slots=1 << 21)
    whole.count_bases(reads)
    assert a.lookup(qs) == whole.lookup(qs)
    for t in (a, b, whole):
        t.close()


def test_fit_cuts_an_oversized_table_to_size(KT):
    k = 37
    _, reads, asm = workload(11, 100_000, k)
    t = KT(k, min_slots=1 << 24)
    t.count_bases(reads)
    qs = [asm[i:i + k] for i in range(0, 20_000, 3)]
    want, h, d = t.lookup(qs), t.histogram(), t.info()["distinct"]
    t.fit(0.5)
    slots = t.info()["slots"]
    assert slots < (1 << 24) and d <= 0.5 * slots and (d > 0.25 * slots or slots == 1 << 21)
    assert t.info()["distinct"] == d and t.lookup(qs) == want and t.histogram() == h
    t.fit(0.5)                                    # already fitting: nothing moves
    assert t.info()["slots"] == slots
    t.close()


def test_counting_file_ranges_adds_up_to_the_whole_file(KT, tmp_path):
    """jasper_count_reads_file_ranges over the cuts of dist.plan_read_shards == jasper_count_reads_files over the file"""
    import gzip
    from jasper_amd import dist as jd
    k = 37
    rng = np.random.default_rng(3)
    genome = synth.make_genome(rng, 80_000)
    reads = synth.make_reads_stream(rng, genome, 25, 150, 0.003).tobytes().split(b"N")
    fq = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, r, b"@" * len(r)) for i, r in enumerate(reads) if r)
    p = tmp_path / "reads.fq"
    p.write_bytes(fq)
    fa = tmp_path / "more.fq"
    fa.write_bytes(fq[: fq.index(b"\n@r5000\n") + 1])
    whole = KT(k, min_slots=1 << 21)
    whole.count_files([str(p), str(fa)])
    for world in (2, 5):
        t = KT(k, min_slots=1 << 21)
        for shard in jd.plan_read_shards([str(p), str(fa)], world):
            assert shard, "every rank gets a piece of files this large"
            t.count_file_ranges(shard)
        assert t.info()["occurrences"] == whole.info()["occurrences"] and t.info()["distinct"] == whole.info()["distinct"]
        assert t.histogram() == whole.histogram()
        t.close()
    # a gzip file cannot be cut: read whole by the reader whose range starts at 0, skipped by the others
    gz = tmp_path / "reads.fq.gz"
    with gzip.open(gz, "wb") as f:
        f.write(fq)
    a, b = KT(k, min_slots=1 << 21), KT(k, min_slots=1 << 21)
    a.count_file_ranges([(str(gz), 0, -1), (str(fa), 0, -1)])
    b.count_file_ranges([(str(gz), 1000, -1), (str(fa), 0, 0)])
    assert a.histogram() == whole.histogram() and b.info()["occurrences"] == 0
    for t in (a, b, whole):
        t.close()


# ---- two processes, IPC-mapped peers -------------------------------------------------------------------------------
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
    from jasper_amd import KmerTable, dist as jd
    di = 0 if one_gpu else rank                          # one rank per device when the box has them (backend nccl = RCCL)
    if backend == "nccl":
        torch.cuda.set_device(di)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", di))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        k = 37
        rng = np.random.default_rng(43)                     # same workload on every rank
        genome = synth.make_genome(rng, 150_000)
        reads = synth.make_reads_stream(rng, genome, 30, 150, 0.003)
        asm = synth.make_assembly(rng, genome, err=1e-3, n_every=10**9).tobytes().decode()
        nrec = reads.size // 151
        dev = torch.device("cuda", di)
        local = KmerTable(k, min_slots=1 << 21, device=di)
        shard = KmerTable(k, min_slots=1 << 16, device=di)
        out = []
        for step in range(2):                               # second round: other read shards, same shard tables reused
            lo, hi = jd.shard_range(nrec, (rank + step) % world, world)
            local.clear()
            local.count_bases(reads[lo * 151:hi * 151].tobytes())
            got = jd.shard_tables(local, shard, dev)
            h = jd.histogram_sharded(shard, dev)
            bs = 20_000
            recs = synth.chunk_records("c", len(asm), bs)
            owner = jd.assign_chunks([b - a for _, a, b in recs], world)
            my = [i for i, o in enumerate(owner) if o == rank]
            res = shard.polish_batch([asm[recs[i][1]:recs[i][2]] for i in my], 3, 2)
            qv = jd.all_reduce_ints(list(res.qv), device=dev)
            out.append((h, shard.info()["distinct"], got, my, res.seqs, qv, shard.info()["slots"]))
            dist.barrier()                                  # (peers may still be reading my shard)
        q.put((rank, out))
        dist.barrier()
        shard.close()
        local.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu_ipc_shards(hip):
    import torch.multiprocessing as mp
    from jasper_amd import KmerTable
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
        assert r0[0] == h and r1[0] == h                          # owners' histograms summed == whole table's
        assert r0[1] + r1[1] == t.info()["distinct"]              # the owners' key sets are a disjoint cover
        assert abs(r0[1] - r1[1]) < 0.05 * t.info()["distinct"]
        assert r0[6] == r1[6]                                     # one geometry
        got = [None] * len(recs)
        for r in (r0, r1):
            for i, s in zip(r[3], r[4]):
                got[i] = s
        assert got == full.seqs                                   # chunk shards through IPC-mapped owners == unsharded run
        assert tuple(r0[5]) == tuple(r1[5]) == full.qv
    t.close()


def test_ipc_mapping_of_a_2_gib_slot_array(hip):
    """hipIpcOpenMemHandle hangs on an allocation of exactly 2^31 bytes on this stack (found rehearsing N=4): a table of 2^27
    slots must still be mappable by its peers (table.hip: slot_alloc_bytes).  Two processes, a hard time limit."""
    import subprocess
    env = dict(os.environ, PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "tools", "ipc_probe.py"), "27", "one_gpu"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert p.stdout.count("attached 1 peers of 2^27 slots") == 2 and p.stdout.count("lookup through the sharded view -> [1]") == 2


def _probe_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      JASPER_NO_IPC_PAD="1", JASPER_AMD_IPC_PROBE_SECONDS="15")
    import time
    import torch
    import torch.distributed as dist
    from jasper_amd import KmerTable, dist as jd
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        local = KmerTable(37, min_slots=1 << 21, device=0)
        local.count_bases(b"ACGTTGCATGCAAGTCCGATAGGCTAACGTTTGACCATGACAGATTACAGGCATCGATCGGATC" if rank == 0 else b"TTGACCATGACAGATTACAGGCATCGATCGGATCAAGGTTCCAAGGTTACGTAGCTAGCTAGGA")
        shard = KmerTable(37, min_slots=1 << 27, device=0)      # 2^31 bytes, NOT padded (JASPER_NO_IPC_PAD): the size that hangs
        shard._fitted = True                                     # (keep that size)
        t0 = time.time()
        try:
            jd.shard_tables(local, shard, dev)
            out = "attached"
        except jd.ShardAttachError as e:
            out = "refused: " + str(e)
        q.put((rank, out, time.time() - t0))
        dist.barrier()
        shard.close()
        local.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(not os.environ.get("JASPER_TEST_IPC_HANG"), reason="provokes a hung driver call in a helper process on purpose: on request only (JASPER_TEST_IPC_HANG=1)")
def test_a_mapping_that_never_returns_is_caught_by_the_probe(hip):
    """without the padding of slot_alloc_bytes a peer's 2 GiB slot array cannot be mapped -- hipIpcOpenMemHandle never returns.
    The throw-away probe process takes that hit: both ranks get ShardAttachError together, within the probe's time limit,
    and can go on (bench.py / cli.py then replicate the table instead)"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_probe_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=200) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1].startswith("refused: ") and "probe process" in r[1] for r in res), res
    assert all(r[2] < 120 for r in res), res
