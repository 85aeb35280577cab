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
