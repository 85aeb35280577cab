"""GPU: the same input, many times, must give the same bytes.

The product hands data from wave to wave and from kernel to kernel through device memory in several hand-rolled ways (DESIGN.md
5, "hand-offs"): the path-search scratch pool of the polisher (a slot is given up behind an agent-scope release and taken with an
acquire), the arrival flags of chained segments, the shared slice cursors of the counting passes (returning agent-scope
atomics), the deferred lists.  A mistake in any of them shows as a result that changes from run to run -- round 3 found one
(a workgroup-scope fence, `302d64e`) only because a test happened to be flaky.  Here the bench workload (BASELINE configs[1]
shape: 47 Mb, 30x, k = 37, 2 passes, chunked as `jasper.sh -t 16`; src/jasper.py:527-583 base_extension is the code that race
corrupted) is run 20 times in each configuration, with few scratch slots, one and three lanes, and while another stream
saturates the HBM, and every run has to reproduce the first one's polished text, fix records, QV counters and table exactly."""
import hashlib
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GENOME_MB = float(os.environ.get("JASPER_TEST_DETERMINISM_MB", "47"))
RUNS = 20


class HbmNoise:
    """copies between two 1-GiB buffers on a side stream: ~30 ms of memory traffic per burst, enqueued right before a call"""

    def __init__(self, torch, dev):
        self.torch = torch
        self.stream = torch.cuda.Stream(device=dev)
        self.a = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        self.b = torch.empty(1 << 30, dtype=torch.uint8, device=dev)

    def burst(self, n=48):
        with self.torch.cuda.stream(self.stream):
            for _ in range(n // 2):
                self.b.copy_(self.a, non_blocking=True)
                self.a.copy_(self.b, non_blocking=True)

    def drain(self):
        self.stream.synchronize()


@pytest.fixture(scope="module")
def work(hip):
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from jasper_amd import KmerTable
    dev = torch.device("cuda", 0)
    reads, names, seqs, d_chunks, asm_len, bs, nreads = bench.build_workload(torch, dev, 0, 1, GENOME_MB, 2)
    jf_size = int(nreads * bench.READ_LEN * 2.1 / 10)
    t = KmerTable(bench.K, min_slots=max(1 << 21, int(1.25 * jf_size)))
    noise = HbmNoise(torch, dev)
    yield dict(torch=torch, reads=reads, d_chunks=d_chunks, table=t, noise=noise, n=len(seqs))
    t.close()


def _count_digest(torch, t, reads, buf):
    """histogram + an order-independent digest of every (hash, count) entry of the table, taken on the device"""
    t.clear()
    t.count_bases_device(reads.data_ptr(), reads.numel())
    h = t.histogram()
    info = t.info()
    n = t.export_to(buf.data_ptr(), buf.numel() // 3)           # [n, 3] int64: mixed-hash hi, lo, count
    assert n == info["distinct"]
    e = buf[: 3 * n].view(n, 3)
    row = (e[:, 0] * -7046029254386353131 + e[:, 1]) * -4417276706812531889 + e[:, 2] * 1442695040888963407     # (wraps: arithmetic mod 2^64)
    row = row ^ (row >> 29)
    return (hashlib.sha256(repr(h).encode()).hexdigest(), int(row.sum().item()), int((row * row).sum().item()), int(e[:, 2].sum().item()),
            info["distinct"], info["occurrences"])


def test_counting_the_same_reads_twenty_times(work):
    torch, t, reads, noise = work["torch"], work["table"], work["reads"], work["noise"]
    t.clear()
    t.count_bases_device(reads.data_ptr(), reads.numel())
    buf = torch.empty(3 * (t.info()["distinct"] + 1024), dtype=torch.int64, device=reads.device)
    want = _count_digest(torch, t, reads, buf)
    assert t.count_path() == 1                  # the partition passes (shared slice cursors, deferred lists), not the direct kernel
    assert want[3] == want[5]                   # (every occurrence is in some entry's count)
    for it in range(RUNS - 1):
        if it % 2 == 0:
            noise.burst()
        assert _count_digest(torch, t, reads, buf) == want, it
    noise.drain()


def _polish_digest(t, d_chunks, thr, n):
    r = t.polish_batch_device(d_chunks[0], d_chunks[1], thr, 2, fix=True)
    h = hashlib.sha256()
    for i in range(n):
        h.update(bytes(r.seq_view(i)))
    h.update(r._raw.tobytes())
    for a in r.aux:
        h.update(a)
    return h.hexdigest(), r.qv, r.n_records, r.lookups


@pytest.mark.parametrize("lanes,nslots,busy", [(1, 8, False), (1, 8, True), (3, 8, True), (1, 0, True), (0, 0, True)])
def test_polishing_the_same_batch_twenty_times(work, lanes, nslots, busy):
    from jasper_amd import polisher
    t, noise = work["table"], work["noise"]
    t.clear()
    t.count_bases_device(work["reads"].data_ptr(), work["reads"].numel())
    txt, status = polisher.threshold_from_histo_rows(t.histo_rows())
    assert status == 0 and txt
    thr = int(txt)
    for v in ("JASPER_POLISH_LANES", "JASPER_POLISH_TEST_NSLOTS"):
        os.environ.pop(v, None)
    os.environ["JASPER_POLISH_LANES"] = "1"
    want = _polish_digest(t, work["d_chunks"], thr, work["n"])        # 256 slots, one lane, idle memory
    os.environ.pop("JASPER_POLISH_LANES")
    if lanes:                                                         # (0: the product's own choice -- one lane on a table's first polishing call, three from the second on)
        os.environ["JASPER_POLISH_LANES"] = str(lanes)
    if nslots:
        os.environ["JASPER_POLISH_TEST_NSLOTS"] = str(nslots)
    try:
        for it in range(RUNS):
            if busy:
                noise.burst()
            assert _polish_digest(t, work["d_chunks"], thr, work["n"]) == want, (lanes, nslots, busy, it)
    finally:
        noise.drain()
        for v in ("JASPER_POLISH_LANES", "JASPER_POLISH_TEST_NSLOTS"):
            os.environ.pop(v, None)
