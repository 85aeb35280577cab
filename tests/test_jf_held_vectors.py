"""The golden vectors the reference itself HOLDS on the counting path: Jellyfish's own test suite pins the md5 of what
`jellyfish histo` prints after `jellyfish count -C -m 15` on inputs made by its own seeded generator

    JF::tests/parallel_hashing.sh:6-20   864c0b0826854bdc72a85d170549b64b   seq10m.fa (-s 2M with doubling, -s 16M, and DOS line ends)
    JF::tests/multi_file.sh:6-16         d93b7678037814c256d1d9120a0e6422   seq1m_0 seq1m_1 seq1m_2 seq10m seq1m_3 seq1m_4 (plain;
                                                                            and seq10m.fa + the five files gzipped)

The inputs (tests/golden/jf_tests/, made by tests/golden/make_jf_test_vectors.py from the outputs of the reference's
`generate_sequence` with the seeds of JF::tests/generate_sequence.sh:6-7) are multi-line FASTA of 70 columns.  The CPU tests
take the oracle through them, the `-m gpu` tests the HIP path (files -> HBM table -> histogram); both format the histogram as
`jellyfish histo` does (JF::sub_commands/histo_main.cc:82-84: "<multiplicity> <distinct>\\n", empty rows left out) and must
reproduce the md5 the reference's scripts hold."""
import gzip
import hashlib
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "golden", "jf_tests")
MD5_SEQ10M = "864c0b0826854bdc72a85d170549b64b"        # JF::tests/parallel_hashing.sh:7-8,20
MD5_MULTI = "d93b7678037814c256d1d9120a0e6422"         # JF::tests/multi_file.sh:7-8
MULTI_ORDER = ["seq1m_0.fa", "seq1m_1.fa", "seq1m_2.fa", "seq10m.fa", "seq1m_3.fa", "seq1m_4.fa"]      # JF::tests/multi_file.sh:12
K = 15


def _plain(name):
    return gzip.open(os.path.join(DATA, name + ".gz"), "rb").read()


def _histo_md5(rows):
    return hashlib.md5("".join("%d %d\n" % (m, n) for m, n in rows).encode()).hexdigest()


def test_fixture_files_are_the_generators_outputs():
    for ln in open(os.path.join(DATA, "inputs.md5")):
        want, name = ln.split()
        assert hashlib.md5(_plain(name)).hexdigest() == want, name


def _oracle_rows(texts):
    from oracle import oracle as O
    db = O.OracleDB(K)
    for t in texts:
        db.count_text(t)
    h = db.histo()
    return [(m, h[m]) for m in range(1, 10002) if h[m]]


def test_oracle_reproduces_parallel_hashing_md5():
    assert _histo_md5(_oracle_rows([_plain("seq10m.fa")])) == MD5_SEQ10M


def test_oracle_reproduces_parallel_hashing_md5_with_dos_line_ends():
    """`unix2dos -n seq10m.fa seq10mDOS.fa` (JF::tests/generate_sequence.sh:22-24), same md5 (JF::tests/parallel_hashing.sh:20,82-88)"""
    dos = _plain("seq10m.fa").replace(b"\n", b"\r\n")
    assert _histo_md5(_oracle_rows([dos])) == MD5_SEQ10M


def test_oracle_reproduces_multi_file_md5():
    assert _histo_md5(_oracle_rows([_plain(n) for n in MULTI_ORDER])) == MD5_MULTI


# ---- the HIP path: files -> table -> histogram ---------------------------------------------------------------------------

def _write_plain(tmp_path, name, dos=False):
    p = os.path.join(str(tmp_path), name if not dos else name.replace(".fa", "DOS.fa"))
    raw = _plain(name)
    with open(p, "wb") as f:
        f.write(raw.replace(b"\n", b"\r\n") if dos else raw)
    return p


def _gpu_rows(paths, min_slots):
    from jasper_amd.table import KmerTable
    t = KmerTable(K, min_slots=min_slots)
    try:
        t.count_files(paths)
        return t.histo_rows()
    finally:
        t.close()


@pytest.mark.gpu
@pytest.mark.parametrize("min_slots", [2 << 20, 16 << 20])       # `-s 2M` (the table doubles on the way, 9.9 M distinct keys) and `-s 16M`
def test_gpu_reproduces_parallel_hashing_md5(hip, tmp_path, min_slots):
    assert _histo_md5(_gpu_rows([_write_plain(tmp_path, "seq10m.fa")], min_slots)) == MD5_SEQ10M


@pytest.mark.gpu
def test_gpu_reproduces_parallel_hashing_md5_with_dos_line_ends(hip, tmp_path):
    assert _histo_md5(_gpu_rows([_write_plain(tmp_path, "seq10m.fa", dos=True)], 2 << 20)) == MD5_SEQ10M


@pytest.mark.gpu
def test_gpu_reproduces_multi_file_md5_plain_and_gzipped(hip, tmp_path):
    plain = [_write_plain(tmp_path, n) for n in MULTI_ORDER]
    assert _histo_md5(_gpu_rows(plain, 2 << 20)) == MD5_MULTI
    # JF::tests/multi_file.sh:21-23: seq10m.fa as a file, the five others through `gunzip -c` generators -- here as .gz files, which
    # the reader inflates itself (role of `zcat -f`, src/jasper.sh:177)
    zipped = [os.path.join(DATA, "seq1m_%d.fa.gz" % i) for i in (1, 3, 4, 2, 0)] + [plain[3]]
    assert _histo_md5(_gpu_rows(zipped, 2 << 20)) == MD5_MULTI


@pytest.mark.gpu
def test_gpu_text_entry_point_reproduces_the_md5s(hip):
    """the same through jasper_count_reads_text (an in-memory stream), several calls into one table"""
    from jasper_amd.table import KmerTable
    t = KmerTable(K, min_slots=2 << 20)
    try:
        for n in MULTI_ORDER:
            t.count_text(_plain(n))
        assert _histo_md5(t.histo_rows()) == MD5_MULTI
    finally:
        t.close()
