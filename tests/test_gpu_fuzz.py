"""GPU: randomised differential test HIP path vs oracle, with the case generator that pinned the oracle against the real
reference (tests/golden/fuzz_vs_reference.py: homopolymers, tandem repeats, diploid sites, high-copy elements, coverage gaps,
sub/ins/del/lower-case/N/IUPAC edits, every read-file format, random k / passes / threshold / chunking)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def _one(seed, KmerTable, polisher, O, G, F, tmp_path):
    rng, spec, haps, chunks = F.random_case(seed)
    k = spec["k"]
    reads = G.sample_reads(rng, haps, spec["cov"], spec["rl"], spec["err"])
    if not reads:
        reads = [haps[0][0][:spec["rl"]]]
    ext = "fq" if spec["fmt"].startswith("fq") else "fa"
    rpath = str(tmp_path / ("reads%d.%s" % (seed, ext)))
    G.write_reads(rpath, reads, spec["fmt"], rng)
    text = open(rpath, "rb").read()
    odb = O.OracleDB(k)
    odb.count_text(text)
    t = KmerTable(k, min_slots=1 << 16)
    t.count_files([rpath])
    items = list(odb.items())
    assert t.info()["distinct"] == len(items) and t.histogram() == odb.histo(), (seed, spec)
    sample = items[:: max(1, len(items) // 200)]
    assert t.lookup([km for km, _ in sample]) == [min(c, 0xFFFFFFFF) for _, c in sample], (seed, spec)
    names = [c[0] for c in chunks]
    seqs = [c[1] for c in chunks]
    try:
        fixed_o, rows_o, qv_o, _ = odb.polish_batch(names, seqs, spec["thre"], spec["passes"])
        ok_o = True
    except RuntimeError:
        ok_o = False
    try:
        fixed, rows, qv, _ = polisher.polish_batch(t, names, seqs, spec["thre"], spec["passes"])
        ok = True
    except Exception:
        ok = False
    assert ok == ok_o, (seed, spec)
    if ok:
        assert qv == qv_o and fixed == fixed_o, (seed, spec)
        for it in range(spec["passes"]):
            assert polisher.fix_csv_text(rows[it]) == "Contig Base_coord Original Mutation\r\n" + rows_o[it], (seed, spec, it)
    t.close()


def test_fuzz_hip_vs_oracle(hip, tmp_path):
    from jasper_amd import KmerTable, polisher
    from oracle import oracle as O
    import make_golden as G              # generator helpers only; nothing of the reference is touched
    import fuzz_vs_reference as F
    for seed in range(5000, 5300):
        _one(seed, KmerTable, polisher, O, G, F, tmp_path)


def test_fuzz_reads_through_owner_shards(hip, tmp_path):
    """the same cases with every read (histogram, lookups, polishing) going through 2..5 owner tables split off the counted
    one (tools/fuzz_shard.py): an owner-sharded table must be indistinguishable from a whole one"""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    from jasper_amd import KmerTable, polisher
    from oracle import oracle as O
    import make_golden as G
    import fuzz_vs_reference as F
    from fuzz_shard import sharded_class
    S = sharded_class(KmerTable)
    for seed in range(9000, 9060):
        _one(seed, S, polisher, O, G, F, tmp_path)


def test_fuzz_counting_by_list_exchange(hip, tmp_path):
    """the same cases counted the multi-GPU way on the one GPU (tools/fuzz_exchange.py): read feed -> every batch cut into
    2..5 ranges wherever the cuts fall -> region lists by key owner -> owners' shards; all reads through the shards"""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    from jasper_amd import KmerTable, polisher
    from oracle import oracle as O
    import make_golden as G
    import fuzz_vs_reference as F
    from fuzz_exchange import exchanged_class
    S = exchanged_class(KmerTable)
    for seed in range(9500, 9560):
        _one(seed, S, polisher, O, G, F, tmp_path)
    assert S.taken[0] >= 30                         # (most cases have a table with an exchange geometry)


def test_fuzz_wide_k(hip, tmp_path, monkeypatch):
    """the same generator with k in {38, 45, 51, 57, 63} mixed in (tables of few slots: remainders wider than the tag word);
    the oracle was checked against the real reference on these k too (tests/golden/fuzz_vs_reference.py with
    JASPER_FUZZ_WIDE_K=1: 750 cases, 0 failing)"""
    monkeypatch.setenv("JASPER_FUZZ_WIDE_K", "1")
    from jasper_amd import KmerTable, polisher
    from oracle import oracle as O
    import make_golden as G
    import fuzz_vs_reference as F
    wide = 0
    for seed in range(12000, 12120):
        wide += F.random_case(seed)[1]["k"] >= 38
        _one(seed, KmerTable, polisher, O, G, F, tmp_path)
    assert wide >= 20


def test_the_same_small_cases_over_and_over(hip, tmp_path):
    """A result that depends on timing shows only when the same case runs many times; tools/fuzz_repeat.py is the long form of this
    test.  A few cases with path searches, chained segments and the host parser, 40 times each.  (Round 5's open observation -- seed
    995395 differing from the oracle about once in 1 000 - 20 000 repeats on some boxes of the pool, with two experimental builds of
    the walk -- is described in DESIGN.md 2 and docs/experiments.md; that seed is not part of this test, whose job is to be a
    deterministic regression check.)"""
    from jasper_amd import KmerTable, polisher
    from oracle import oracle as O
    import make_golden as G
    import fuzz_vs_reference as F
    for rep in range(40):
        for seed in (995393, 995396, 5007, 5131):
            _one(seed, KmerTable, polisher, O, G, F, tmp_path)
            for f in tmp_path.iterdir():
                f.unlink()
