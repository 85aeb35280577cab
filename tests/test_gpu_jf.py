"""GPU: reading the reference's own database files (`jellyfish count` output, tests/golden/cases/*/db.jf) -- the path
behind `jasper.sh -j` and behind jf.QueryMerFile(path) (JF::swig/mer_file.i:18-41)."""
import os
import shutil
import subprocess
import sys

import pytest

from golden_util import Case, fasta60

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["simple_k25", "simple_k37"])
def test_load_jf_equals_reference_dump_and_polishes(hip, name):
    from jasper_amd import KmerTable, polisher
    c = Case(name)
    t = KmerTable.from_jf(os.path.join(c.dir, "db.jf"))
    assert t.k == c.k                               # k comes from the header
    d = c.dump()
    kmers = sorted(d)
    assert t.lookup(kmers) == [d[x] for x in kmers]
    assert t.info()["distinct"] == len(d)
    assert t.histo_rows() == c.histo_rows()
    # the reference's own flow: jasper.py --db db.jf --query batch.fa ...
    names, seqs = c.batch()
    fixed, rows, qv, _ = polisher.polish_batch(t, names, seqs, c.thre, c.passes)
    assert qv == c.qv() and fasta60(names, fixed) == c.fixed_fa()
    for it in range(c.passes):
        assert polisher.fix_csv_text(rows[it]) == c.fix_csv(it)
    t.close()


def test_load_jf_errors(hip, tmp_path):
    from jasper_amd import KmerTable
    from jasper_amd._lib import JasperHipError
    with pytest.raises(JasperHipError, match="Can't open file"):
        KmerTable.from_jf(str(tmp_path / "missing.jf"))
    p = tmp_path / "bad.jf"
    p.write_bytes(b"not a jellyfish file at all")
    with pytest.raises(JasperHipError, match="Unsupported format"):
        KmerTable.from_jf(str(p))


def test_cli_with_jf_database(hip, tmp_path):
    """jasper.sh -j DB -a asm: threshold from the DB's histogram, polishing against the DB"""
    c = Case("simple_k25")
    shutil.copy(os.path.join(c.dir, "db.jf"), tmp_path / "db.jf")
    names, seqs = c.batch()
    with open(tmp_path / "asm.fa", "w") as f:
        f.write(">ctg1\n%s\n" % seqs[0])
    env = dict(os.environ, PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, "-m", "jasper_amd.cli", "-j", "db.jf", "-a", "asm.fa", "-k", "31", "-t", "1", "-p", "2", "-b", "1000000"],
                       cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "Computing K-mer histogram" in p.stdout
    assert open(tmp_path / "jfhisto31.csv").read() == open(os.path.join(c.dir, "histo.csv")).read()   # named after -k, content from the DB
    polished = open(tmp_path / "asm.fa.polished.fasta").read().split("\n")
    assert polished[0] == ">ctg1" and len(polished[1]) > 3900
