"""CPU: dist.plan_read_shards -- where the read files are cut for the GPUs of one node.  A cut is only correct at a record
start (k-mers never span records, counts are sums over reads), and input that only makes sense as ONE stream
(src/jasper.sh:177 `zcat -f $READS`) must not be cut at all."""
import gzip

import numpy as np
import pytest

from jasper_amd import dist as jd


def _fastq(rng, n, lo=30, hi=151):
    out = []
    for i in range(n):
        L = int(rng.integers(lo, hi))
        seq = "".join(rng.choice(list("ACGTN"), L, p=[.24, .24, .24, .24, .04]))
        q = "".join(rng.choice(list("@I+#>"), L))            # quality lines that look like headers / separators
        out.append("@r%d/1 x\n%s\n+\n%s\n" % (i, seq, q))
    return "".join(out)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_fastq_cuts_are_record_starts_and_cover_the_file(tmp_path, world):
    rng = np.random.default_rng(world)
    txt = _fastq(rng, 4000).encode()
    p = tmp_path / "a.fq"
    p.write_bytes(txt)
    sh = jd.plan_read_shards([str(p)], world)
    assert len(sh) == world
    pieces = sorted((b, e) for r in sh for _, b, e in r)
    assert pieces[0][0] == 0 and pieces[-1][1] == len(txt)
    for (b0, e0), (b1, e1) in zip(pieces, pieces[1:]):
        assert e0 == b1                                       # disjoint cover, in order
    for b, e in pieces:
        assert txt[:b].count(b"\n") % 4 == 0 and txt[b:b + 1] == b"@"
    sizes = [e - b for b, e in pieces]
    assert max(sizes) < 1.2 * len(txt) / world + 1000         # balanced


def test_fasta_cuts_and_tiny_files(tmp_path):
    rng = np.random.default_rng(5)
    fa = "".join(">s%d\n%s\n%s\n" % (i, "".join(rng.choice(list("ACGT"), 70)), "".join(rng.choice(list("ACGT"), 33))) for i in range(500)).encode()
    p = tmp_path / "r.fa"
    p.write_bytes(fa)
    sh = jd.plan_read_shards([str(p)], 4)
    pieces = sorted((b, e) for r in sh for _, b, e in r)
    assert pieces[0][0] == 0 and pieces[-1][1] == len(fa) and all(fa[b:b + 1] == b">" for b, _ in pieces)
    assert all(a[1] == b[0] for a, b in zip(pieces, pieces[1:]))
    # fewer records than ranks: some ranks get nothing, nothing is lost or doubled
    q = tmp_path / "one.fq"
    q.write_bytes(b"@a\nACGT\n+\nIIII\n")
    sh = jd.plan_read_shards([str(q)], 4)
    assert sorted((b, e) for r in sh for _, b, e in r) == [(0, 15)]


def test_input_that_is_one_stream_is_not_cut(tmp_path):
    rng = np.random.default_rng(6)
    fq = tmp_path / "a.fq"
    fq.write_bytes(_fastq(rng, 300).encode())
    fa = tmp_path / "b.fa"
    fa.write_bytes(b">x\nACGTACGT\n")
    whole = lambda paths: [[(str(p), 0, -1) for p in paths], []]
    assert jd.plan_read_shards([str(fq), str(fa)], 2) == whole([fq, fa])            # formats differ: the first byte of the STREAM decides
    nonl = tmp_path / "c.fq"
    nonl.write_bytes(_fastq(rng, 50).encode().rstrip(b"\n"))
    assert jd.plan_read_shards([str(nonl), str(fq)], 2) == whole([nonl, fq])         # the next file would continue c.fq's last line
    # multi-line FASTQ has no 4-line record starts: not cut blindly
    ml = tmp_path / "ml.fq"
    ml.write_bytes(("".join("@m%d\nACGTACGTAC\nGTACGTACGT\n+\nIIIIIIIIII\nIIIIIIIIII\n" % i for i in range(3000))).encode())
    sh = jd.plan_read_shards([str(ml)], 2)
    assert sorted((b, e) for r in sh for _, b, e in r) in ([(0, -1)], [(0, ml.stat().st_size)])


def test_gzip_files_go_whole_to_the_least_loaded_rank(tmp_path):
    rng = np.random.default_rng(7)
    paths = []
    for i, n in enumerate((900, 300, 250)):
        p = tmp_path / ("g%d.fq.gz" % i)
        with gzip.open(p, "wb") as f:
            f.write(_fastq(rng, n).encode())
        paths.append(str(p))
    sh = jd.plan_read_shards(paths, 2)
    assert sorted(x for r in sh for x in r) == sorted((p, 0, -1) for p in paths)
    assert [p for p, _, _ in sh[0]] == [paths[0]] and sorted(p for p, _, _ in sh[1]) == sorted(paths[1:])


def test_pipes_are_not_touched(tmp_path):
    import os
    fifo = tmp_path / "r.fifo"
    os.mkfifo(fifo)
    plain = tmp_path / "a.fq"
    plain.write_bytes(b"@a\nACGT\n+\nIIII\n")
    # (opening the FIFO would block forever here: the plan must come back without looking at it)
    assert jd.plan_read_shards([str(plain), str(fifo)], 2) == [[(str(plain), 0, -1), (str(fifo), 0, -1)], []]


def test_stated_rules_for_dedupe_and_replication(monkeypatch):
    """dist.dedupe_pays / dist.prefer_replicated: the bytes-per-link rules of DESIGN.md section 7 (models: nothing has run over xGMI)"""
    from jasper_amd import dist as jdist
    monkeypatch.delenv("JASPER_AMD_EXCHANGE_DEDUPE", raising=False)
    assert [jdist.dedupe_pays(w) for w in (2, 3, 4, 6, 8)] == [True, True, True, False, False]
    assert jdist.dedupe_pays(8, setting="1") and not jdist.dedupe_pays(2, setting="0")
    hbm = 288e9
    # BASELINE shapes, one polish call of P + 1 scans per counted table: the gather costs more than every remote lookup of the run
    assert not jdist.prefer_replicated(2, 0.5e9, 70e6, 3, 0.9 * hbm)              # configs[2]
    assert not jdist.prefer_replicated(8, 8 * 157e6, 47e6, 3, 0.9 * hbm)          # the bench's weak scaling at N = 8
    assert not jdist.prefer_replicated(8, 8e9, 390e6, 3, 0.9 * hbm)               # configs[3]: 2^34 slots x 16 B x 2 do not even fit
    # a resident table that is polished over and over is worth a copy per GPU -- if it fits
    assert jdist.prefer_replicated(8, 8 * 157e6, 47e6, 3, 0.9 * hbm, polish_calls=200)
    assert not jdist.prefer_replicated(8, 8 * 157e6, 47e6, 3, 30e9, polish_calls=200)
