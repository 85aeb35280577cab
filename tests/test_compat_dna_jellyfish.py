"""The `dna_jellyfish`-shaped module (jasper_amd/compat/dna_jellyfish.py) against the real SWIG binding's answers.

tests/golden/mer_kats.json: what the reference's own `dna_jellyfish.MerDNA(s)` / `.get_canonical()` print (Jellyfish 2.3.0 built
from the vendored tarball, tests/golden/make_golden.py); tests/golden/cases/*/dump.txt.gz: `jellyfish dump -c` of its databases.
The three calls src/jasper.py makes (:15, :70, :71) are the ones that matter; JF::swig/mer_file.i:12-43, mer_dna.i:12-19."""
import json
import os

import pytest

from golden_util import Case

HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "mer_kats.json")))


def test_merdna_strings_equal_the_swig_binding():
    import jasper_amd.compat.dna_jellyfish as jf
    assert len(KATS) >= 150
    for r in KATS:
        jf.MerDNA.k(r["k"])
        m = jf.MerDNA(r["s"])
        assert str(m) == r["mer"], r
        assert str(m.get_canonical()) == r["canonical"], r
        assert jf.MerDNA.k() == r["k"] and len(m) == r["k"]
        c = jf.MerDNA(m)
        c.canonicalize()
        assert c == m.get_canonical() and str(c.get_reverse_complement().get_reverse_complement()) == str(c)


def test_merdna_small_api():
    import jasper_amd.compat.dna_jellyfish as jf
    jf.MerDNA.k(4)
    m = jf.MerDNA("ACGT")
    assert m.shift_left("A") == "A" and str(m) == "CGTA"          # the examples of JF::swig/mer_dna.i's own doc strings
    m = jf.MerDNA("ACGT")
    assert m.shift_right("A") == "T" and str(m) == "AACG"
    assert str(jf.MerDNA()) == "AAAA" and jf.MerDNA().is_homopolymer() and not jf.MerDNA("ACGT").is_homopolymer()
    assert jf.MerDNA("AAAC") < jf.MerDNA("AAAG") and jf.MerDNA("TAAA") > jf.MerDNA("GTTT")
    m.polyT()
    assert str(m) == "TTTT"


def test_open_a_file_that_is_not_there():
    """JF::swig/mer_file.i:19-21 -- no GPU is touched before the file has been opened"""
    import jasper_amd.compat.dna_jellyfish as jf
    with pytest.raises(RuntimeError, match=r"^Can't open file '/nonexistent/db\.jf'$"):
        jf.QueryMerFile("/nonexistent/db.jf")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["simple_k25", "simple_k37", "simple_k45"])
def test_query_mer_file_equals_jellyfish_dump(hip, name):
    """jf.QueryMerFile(db)[jf.MerDNA(s).get_canonical()] for every k-mer of the reference's dump, either strand, and for absent ones"""
    import jasper_amd.compat.dna_jellyfish as jf
    c = Case(name)
    db = os.path.join(c.dir, "db.jf")
    if os.path.exists(db):
        qf = jf.QueryMerFile(db)
    else:                                           # (cases without a database file: the counts of the case's reads)
        from jasper_amd import KmerTable
        t = KmerTable(c.k, min_slots=1 << 16)
        t.count_text(c.reads_text())
        qf = jf.QueryMerFile(t)
    assert jf.MerDNA.k() == c.k                    # opening the DB set the process-wide k (mer_file.i:23)
    d = c.dump()
    kmers = sorted(d)[:3000]
    for s in kmers[:40]:                            # the reference's own call shape, one lookup at a time
        assert qf[jf.MerDNA(s).get_canonical()] == d[s]
        assert qf[jf.MerDNA(s.lower()).get_reverse_complement().get_canonical()] == d[s]
    assert qf.counts([jf.MerDNA(s) for s in kmers]) == [d[s] for s in kmers]
    absent = [s for s in (("ACGT" * 16)[:c.k], ("TTGCA" * 13)[:c.k]) if str(jf.MerDNA(s).get_canonical()) not in d]
    assert qf.counts([jf.MerDNA(s).get_canonical() for s in absent]) == [0] * len(absent)
    # truncate-and-A-pad (H2): a window with an N counts as its prefix + poly-A
    s = kmers[0][:10] + "N" + kmers[0][11:]
    padded = kmers[0][:10] + "A" * (c.k - 10)
    assert qf[jf.MerDNA(s).get_canonical()] == d.get(str(jf.MerDNA(padded).get_canonical()), 0)


@pytest.mark.gpu
def test_query_mer_file_unsupported_format(hip, tmp_path):
    import jasper_amd.compat.dna_jellyfish as jf
    p = tmp_path / "bad.jf"
    p.write_bytes(b"not a jellyfish file at all")
    with pytest.raises(RuntimeError, match=r"^Unsupported format '"):
        jf.QueryMerFile(str(p))
    # a header of another Jellyfish format names it, as mer_file.i:34 does
    hdr = json.dumps({"format": "text/sorted", "key_len": 50}).encode()
    hdr += b"\0" * ((-(9 + len(hdr))) % 8)
    p.write_bytes(b"%09d" % len(hdr) + hdr)
    with pytest.raises(RuntimeError, match=r"^Unsupported format 'text/sorted'$"):
        jf.QueryMerFile(str(p))
