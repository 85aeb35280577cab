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
    # the header is JSON with strings the user controls (cmdline, pwd): only its TOP-LEVEL members count
    import json
    c = Case("simple_k25")
    raw = open(os.path.join(c.dir, "db.jf"), "rb").read()
    hlen = int(raw[:9])
    hdr = json.loads(raw[9:9 + hlen].rstrip(b"\0"))
    body = raw[9 + hlen:]

    def write(path, h, data):
        j = json.dumps(h).encode()
        j += b"\0" * ((-(9 + len(j))) % 8)
        path.write_bytes(b"%09d" % len(j) + j + data)
    tricky = {"cmdline": ["count", "-o", '"key_len":10,"format":"text/sorted","counter_len":9'], "pwd": '/tmp/"canonical":false'}
    tricky.update({k: v for k, v in hdr.items() if k not in ("cmdline", "pwd")})
    write(p, tricky, body)
    t = KmerTable.from_jf(str(p))
    d = c.dump()
    kmers = sorted(d)[:500]
    assert t.k == c.k and t.lookup(kmers) == [d[x] for x in kmers]
    t.close()
    other = dict(hdr, format="text/sorted")
    write(p, other, body)
    with pytest.raises(JasperHipError, match="Unsupported format"):
        KmerTable.from_jf(str(p))
    write(p, hdr, body[:-3])                      # not a whole number of records
    with pytest.raises(JasperHipError, match="Unsupported format"):
        KmerTable.from_jf(str(p))
    p.write_bytes(b"999999999{}")                 # a header length beyond the file
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


def _read_jf(path):
    """(header dict, [(key int, count)]) of a binary/sorted file, in file order"""
    import json
    raw = open(path, "rb").read()
    hlen = int(raw[:9])
    hdr = json.loads(raw[9:9 + hlen].rstrip(b"\0").decode())
    kb = (hdr["key_len"] + 7) // 8
    rec = kb + hdr["counter_len"]
    body = raw[9 + hlen:]
    assert (9 + hlen) % 8 == 0 and len(body) % rec == 0
    out = []
    for i in range(0, len(body), rec):
        out.append((int.from_bytes(body[i:i + kb], "little"), int.from_bytes(body[i + kb:i + rec], "little")))
    return hdr, out


def _jf_pos(hdr, key):
    """matrix1 * key & (size-1) as the reader computes it (JF::include/jellyfish/rectangular_binary_matrix.hpp:224-262:
    key bit i selects column c-1-i; identity form = the low r bits)"""
    m = hdr["matrix1"]
    if m.get("identity"):
        return key & ((1 << m["r"]) - 1) & (hdr["size"] - 1)
    res = 0
    for i in range(m["c"]):
        if (key >> i) & 1:
            res ^= m["columns"][m["c"] - 1 - i]
    return res & (hdr["size"] - 1)


@pytest.mark.parametrize("name", ["simple_k25", "simple_k37"])
def test_write_jf_round_trip_and_reader_order(hip, name, tmp_path):
    """jasper_table_write_jf: same records as the reference's own DB of the same reads, in the order its reader needs
    ((pos, key) ascending, pos from the header's matrix), loadable again, and usable by the CLI's reuse path"""
    from jasper_amd import KmerTable
    c = Case(name)
    t = KmerTable(c.k, min_slots=1 << 16)
    t.count_text(c.reads_text())
    out = str(tmp_path / "mer_counts.jf")
    t.write_jf(out, ["count", "-C", "-m", str(c.k)])
    hdr, recs = _read_jf(out)
    ref_hdr, ref_recs = _read_jf(os.path.join(c.dir, "db.jf"))
    assert hdr["format"] == "binary/sorted" and hdr["key_len"] == 2 * c.k and hdr["counter_len"] == 4 and hdr["canonical"] is True
    assert hdr["cmdline"] == ["count", "-C", "-m", str(c.k)] and hdr["size"] == 1 << hdr["matrix1"]["r"]
    assert sorted(recs) == sorted(ref_recs)                                   # same (k-mer, count) set as real jellyfish
    order = [(_jf_pos(hdr, k), k) for k, _ in recs]
    assert order == sorted(order) and len(set(order)) == len(order)
    # the reference file obeys the same rule under ITS matrix (checks that _jf_pos restates the reader correctly)
    ref_order = [(_jf_pos(ref_hdr, k), k) for k, _ in ref_recs]
    assert ref_order == sorted(ref_order)
    t2 = KmerTable.from_jf(out)
    assert t2.k == c.k and t2.histogram() == t.histogram() and t2.info()["distinct"] == len(recs)
    d = c.dump()
    kmers = sorted(d)[:2000]
    assert t2.lookup(kmers) == [d[x] for x in kmers]
    t.close()
    t2.close()


def test_write_jf_small_k_and_empty(hip, tmp_path):
    from jasper_amd import KmerTable
    for k, text in ((15, "ACGTTGCATGCAAGTCCGATAGGCTAACGT" * 3), (32, "ACGTTGCATGCAAGTCCGATAGGCTAACGTTTGACCATGACAGATTACA" * 2), (21, "")):
        t = KmerTable(k, min_slots=1 << 12)
        if text:
            t.count_bases(text)
        p = str(tmp_path / ("k%d.jf" % k))
        t.write_jf(p)
        hdr, recs = _read_jf(p)
        assert len(recs) == t.info()["distinct"]
        order = [(_jf_pos(hdr, key), key) for key, _ in recs]
        assert order == sorted(order)
        t2 = KmerTable.from_jf(p)
        assert t2.histogram() == t.histogram()
        t.close()
        t2.close()


@pytest.mark.parametrize("k", [45, 51, 63, 64])
def test_write_and_load_jf_wide_k(hip, tmp_path, k):
    """k > 43: keys of up to 128 bits (key bytes = ceil(2k/8) up to 16): the file obeys the reader's (pos, key) order, holds
    the same (k-mer, count) set as a Python restatement of `jellyfish count -C`, and loads back into an identical table.
    The files are kept under gpurun_out/ so that the real jellyfish 2.3.0 can read them in the build container
    (tests/golden/check_jf_writer.py)."""
    import collections
    import numpy as np
    from jasper_amd import KmerTable, synth
    rng = np.random.default_rng(k)
    genome = synth.make_genome(rng, 4000, repeat_frac=0)
    reads = synth.make_reads_stream(rng, genome, 8, 130, 0.002).tobytes().decode()
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    want = collections.Counter()
    for r in reads.split("N"):
        for i in range(len(r) - k + 1):
            f = rc = 0
            for ch in r[i:i + k]:
                f = (f << 2) | code[ch]
            for ch in reversed(r[i:i + k]):
                rc = (rc << 2) | (3 - code[ch])
            want[min(f, rc)] += 1
    t = KmerTable(k, min_slots=1 << 12)
    t.count_bases(reads.encode())
    p = str(tmp_path / ("k%d.jf" % k))
    t.write_jf(p, ["count", "-C", "-m", str(k)])
    hdr, recs = _read_jf(p)
    assert hdr["key_len"] == 2 * k
    assert dict(recs) == dict(want) and len(recs) == len(want)
    order = [(_jf_pos(hdr, key), key) for key, _ in recs]
    assert order == sorted(order)
    t2 = KmerTable.from_jf(p)
    assert t2.k == k and t2.histogram() == t.histogram() and t2.info()["distinct"] == len(want)
    g = genome.tobytes().decode()
    qs = [g[i:i + k] for i in range(0, len(g) - k, 53)]
    assert t2.lookup(qs) == t.lookup(qs) and sum(t.lookup(qs)) > 0
    out = os.path.join(ROOT, "gpurun_out", "jf_wide")
    os.makedirs(out, exist_ok=True)
    shutil.copy(p, os.path.join(out, "k%d.jf" % k))
    with open(os.path.join(out, "k%d.reads.fa" % k), "w") as f:
        f.write("".join(">r%d\n%s\n" % (i, r) for i, r in enumerate(reads.split("N")) if r))
    t.close()
    t2.close()
