"""GPU parity against the golden vectors produced by the real reference (tests/golden/cases)."""
import pytest

from golden_util import CSV_HEADER, Case, case_names, fasta60

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tables(hip):
    from jasper_amd import KmerTable
    cache = {}

    def get(name):
        if name not in cache:
            c = Case(name)
            t = KmerTable(c.k, min_slots=1 << 16)
            t.count_text(c.reads_text())
            cache[name] = t
        return cache[name]
    yield get
    for t in cache.values():
        t.close()


@pytest.mark.parametrize("name", case_names())
def test_counts_and_histogram(tables, name):
    c = Case(name)
    t = tables(name)
    d = c.dump()
    kmers = sorted(d)
    got = t.lookup(kmers)
    assert got == [d[x] for x in kmers]
    info = t.info()
    assert info["distinct"] == len(d)
    assert info["occurrences"] == sum(d.values())
    assert t.histo_rows() == c.histo_rows()


@pytest.mark.parametrize("name", case_names())
def test_threshold(tables, name):
    from jasper_amd import polisher
    c = Case(name)
    txt, status = polisher.threshold_from_histo_rows(tables(name).histo_rows())
    assert status == c.meta["jellyfish_py_exit"]
    assert txt == c.meta["jellyfish_py_stdout"]


@pytest.mark.parametrize("name", case_names())
def test_polish(tables, name):
    from jasper_amd import polisher
    c = Case(name)
    names, seqs = c.batch()
    fixed, rows, qv, res = polisher.polish_batch(tables(name), names, seqs, c.thre, c.passes)
    assert qv == c.qv()
    assert fasta60(names, fixed) == c.fixed_fa()
    for it in range(c.passes):
        assert polisher.fix_csv_text(rows[it]) == c.fix_csv(it)
