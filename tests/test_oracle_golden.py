"""CPU: the oracle (oracle/jasper_oracle.c) against the golden vectors produced by the REAL reference
(Jellyfish 2.3.0 + unmodified src/jasper.py + src/jellyfish.py; tests/golden/make_golden.py)."""
import json
import os

import pytest

from golden_util import CSV_HEADER, GOLDEN, Case, case_names, fasta60
from oracle import oracle as O


def decode(k, v):
    return "".join("ACGT"[(v >> (2 * (k - 1 - i))) & 3] for i in range(k))


def test_golden_cases_present():
    assert len(case_names()) >= 15


def test_encoder_known_answers():
    """str(MerDNA(s)) / get_canonical() of the reference's SWIG binding (JF::swig/mer_dna.i:12-19)"""
    kats = json.load(open(os.path.join(os.path.dirname(GOLDEN), "mer_kats.json")))
    assert len(kats) > 100
    for c in kats:
        k, s = c["k"], c["s"]
        taken, v = O.encode(k, s)
        assert decode(k, v) == c["mer"], (k, s)
        assert decode(k, O.canonical(k, v)) == c["canonical"], (k, s)
        assert O.revcomp(k, O.revcomp(k, v)) == v


@pytest.fixture(scope="module")
def dbs():
    cache = {}

    def get(name):
        if name not in cache:
            c = Case(name)
            db = O.OracleDB(c.k)
            db.count_text(c.reads_text())
            cache[name] = db
        return cache[name]
    return get


@pytest.mark.parametrize("name", case_names())
def test_counts_histo_threshold(dbs, name):
    c = Case(name)
    db = dbs(name)
    assert dict(db.items()) == c.dump()              # jellyfish count -C | jellyfish dump -c
    h = db.histo()
    rows = [(m, h[m]) for m in range(1, 10002) if h[m]]
    assert rows == c.histo_rows()                    # jellyfish histo
    try:
        t = O.threshold(rows)
        status = 0
    except SystemExit:
        t, status = None, 1
    assert status == c.meta["jellyfish_py_exit"]     # src/jellyfish.py
    assert (str(t) if t else "") == c.meta["jellyfish_py_stdout"]


@pytest.mark.parametrize("name", case_names())
def test_polish(dbs, name):
    c = Case(name)
    names, seqs = c.batch()
    fixed, rows, qv, nlook = dbs(name).polish_batch(names, seqs, c.thre, c.passes)
    assert qv == c.qv()                              # {0,P}qValCalcHelper.csv
    assert fasta60(names, fixed) == c.fixed_fa()     # _iter{P-1}_*.fixed.fa
    for it in range(c.passes):
        assert CSV_HEADER + rows[it] == c.fix_csv(it)   # _iter{it}_*.fix.csv (CRLF)


def test_query_padding_semantics(dbs):
    """Appendix A.3: MerDNA(str) keeps the prefix up to the first non-ACGT char and fills with 'A'"""
    db = dbs("simple_k25")
    k = 25
    kmer = next(iter(db.items()))[0]
    assert db.query(kmer) == db.query(kmer.lower())
    assert db.query("") == db.query("A" * k)
    assert db.query(kmer[:10]) == db.query(kmer[:10] + "A" * 15)
    assert db.query(kmer[:10] + "N" + kmer[11:]) == db.query(kmer[:10] + "A" * 15)
    assert db.query(kmer + "GGGG") == db.query(kmer)


def test_count_text_errors():
    db = O.OracleDB(21)
    with pytest.raises(RuntimeError, match="Unsupported format"):
        db.count_text("ACGT\n")
    with pytest.raises(RuntimeError, match="Invalid fastq"):
        db.count_text("@r\nACGTACGT\n+\nIIII\n")
    assert O.OracleDB(21).count_text("") == 0


def test_multithreaded_driver_equals_plain_oracle():
    """oracle.OracleDB(k, threads=N) (bench.py's all-core cpu_baseline: reads divided like `jellyfish count -t N`, chunk
    records like `xargs -P N`) gives exactly what the plain single-threaded restatement gives"""
    import numpy as np
    from jasper_amd import synth
    rng = np.random.default_rng(9)
    g = synth.make_genome(rng, 120_000)
    reads = synth.make_reads_stream(rng, g, 25, 100, 0.004)
    asm = synth.make_assembly(rng, g, err=1e-3, n_every=50_000, n_len=40).tobytes().decode()
    for k, threads in ((25, 3), (37, 5)):
        a, b = O.OracleDB(k), O.OracleDB(k, threads=threads)
        assert a.count_bases(reads.tobytes()) == b.count_bases(reads)           # bytes and numpy input
        assert b.count_bases(reads.tobytes()[:5000]) == a.count_bases(reads.tobytes()[:5000])   # a second call adds up
        assert a.distinct() == b.distinct() and a.histo() == b.histo()
        recs = synth.chunk_records("c", len(asm), 17_000)
        names, seqs = [r[0] for r in recs], [asm[x:y] for _, x, y in recs]
        assert a.polish_batch(names, seqs, 3, 2) == b.polish_batch(names, seqs, 3, 2)
        qs = [asm[i:i + k] for i in range(0, 3000, 7)] + ["", "ACGTN"]
        assert [a.query(q) for q in qs] == [b.query(q) for q in qs]


@pytest.mark.skipif(not (os.path.exists("/tmp/jf_install/bin/jellyfish") and os.path.exists("/root/reference/src/jasper.py")),
                    reason="needs the reference built in the build container (SURVEY.md Appendix C)")
def test_generator_reproduces_committed_cases(tmp_path):
    """tests/golden/make_golden.py, run against the REAL reference, must regenerate the committed fixtures byte for byte
    (seeds are kept per case NAME; a seed taken from a case's position changes whenever a case is inserted)"""
    import filecmp
    import subprocess
    import sys
    gen = os.path.join(os.path.dirname(GOLDEN), "make_golden.py")
    for name in ("cluster_k37", "edges_k25"):
        env = dict(os.environ, GOLDEN_KATS="0", GOLDEN_E2E="0")
        subprocess.run([sys.executable, gen, "--out", str(tmp_path), "--only", name], check=True, capture_output=True, env=env, timeout=600)
        new, old = os.path.join(str(tmp_path), name), os.path.join(GOLDEN, name)
        files = sorted(os.listdir(old))
        assert sorted(os.listdir(new)) == files, name
        same, diff, errs = filecmp.cmpfiles(old, new, files, shallow=False)
        assert not diff and not errs, (name, diff, errs)
