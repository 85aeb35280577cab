"""CPU (no GPU call: the jasper_asm_* entry points are host code): the native assembly job -- split, batch files, fixed files,
join -- against the Python restatement of the perl one-liners in jasper_amd/cli.py (src/jasper.sh:132,155-156,220) and of
src/jasper.py:120-128, which tests/test_host_logic.py and the CLI fixtures pin against the real reference."""
import os
import random

import pytest

from jasper_amd import cli, polisher
from jasper_amd.assembly import AssemblyJob


def _fasta(rng, contigs, width, desc=True, blank_lines=False, trailing_newline=True):
    out = []
    for name, n in contigs:
        out.append(">%s%s\n" % (name, (" some description\tx=1" if desc else "")))
        seq = "".join(rng.choice("ACGTNacgtn") for _ in range(n))
        for i in range(0, n, width):
            out.append(seq[i:i + width] + "\n")
            if blank_lines and rng.random() < 0.1:
                out.append("\n")
    text = "".join(out)
    if not trailing_newline and text.endswith("\n"):
        text = text[:-1]
    return text


def _python_split(path, batch_size, prefix, d):
    cwd = os.getcwd()
    os.chdir(d)
    try:
        contigs = cli.read_assembly(path)
        files = cli.split_batches(contigs, batch_size, prefix)
        return contigs, {f: open(f, "rb").read() for f in files}
    finally:
        os.chdir(cwd)


CASES = [
    dict(contigs=[("c1", 1000)], width=60, bs=300),
    dict(contigs=[("c1", 1000), ("c2", 17), ("c3", 2500)], width=70, bs=450),
    dict(contigs=[("a", 5), ("b", 0), ("c", 61), ("d", 60)], width=60, bs=7),                 # an empty contig: no record
    dict(contigs=[("chr%d" % i, 100 + 37 * i) for i in range(40)], width=50, bs=333),
    dict(contigs=[("x", 10000)], width=10**6, bs=1),                                          # one line; records of one base
    dict(contigs=[("x:0", 900), ("x", 901), ("y:7:0", 5)], width=80, bs=225, blank_lines=True),   # names that look like record names
    dict(contigs=[("p", 4096), ("q", 4097)], width=64, bs=4096, trailing_newline=False),
    dict(contigs=[("big", 300000), ("s", 1)], width=60, bs=100000, desc=False),
]


@pytest.mark.parametrize("case", CASES)
def test_native_split_join_and_fixed_files_equal_the_python_rules(tmp_path, case):
    rng = random.Random(len(case["contigs"]) * 1000 + case["bs"])
    text = _fasta(rng, case["contigs"], case["width"], desc=case.get("desc", True), blank_lines=case.get("blank_lines", False),
                  trailing_newline=case.get("trailing_newline", True))
    path = str(tmp_path / "asm.fa")
    open(path, "w").write(text)
    job = AssemblyJob.open(path)
    assert job is not None
    assert job.sequence_bytes == cli.sequence_bytes(path) == cli.sequence_bytes(path, fast=False)
    pd, nd = tmp_path / "py", tmp_path / "native"
    pd.mkdir(), nd.mkdir()
    contigs, want = _python_split(path, case["bs"], "asm.fa", str(pd))
    assert job.contig_names() == [c[0] for c in contigs]
    cwd = os.getcwd()
    os.chdir(nd)
    try:
        n_chunks, n_files = job.split(case["bs"], "asm.fa")
        job.split_wait()
        got = {job.batch_file_name(f): open(job.batch_file_name(f), "rb").read() for f in range(n_files)}
        assert got == want
        assert [len(v) for _, v in sorted(got.items(), key=lambda kv: int(kv[0].split(".")[-2]))] == job.file_bytes
        # what the polisher would read from the batch files
        recs = {}
        for f in range(n_files):
            recs.update(polisher.parse_fasta(job.batch_file_name(f)))
        assert list(recs) == [job.chunk_name(c) for c in range(n_chunks)]
        assert [v.encode() for v in recs.values()] == [job.chunk_text(c) for c in range(n_chunks)]
        # "polished" text: every record changed in length and content (a stand-in for the GPU's result)
        polished = []
        for c in range(n_chunks):
            t = job.chunk_text(c).decode()
            t = (t[: len(t) // 2] + "ACGT"[c % 4] * (c % 5) + t[len(t) // 2 + (c % 3):]) if len(t) > 3 else t
            polished.append(t)
            job.put(c, t)
        # the fixed files (src/jasper.py:120-128) of every batch file, natively and by polisher.main_many's writer
        outs = ["_iter1_%s.fixed.fa" % job.batch_file_name(f) for f in range(n_files)]
        job.write_fixed(list(range(n_files)), outs)
        for f in range(n_files):
            want_fixed = b""
            for c in range(job.file_first[f], job.file_first[f + 1]):
                want_fixed += (">%s\n" % job.chunk_name(c)).encode() + (polisher.wrap_lines(polished[c], 60) or b"")
            assert open(outs[f], "rb").read() == want_fixed
        # the join (src/jasper.sh:220) from memory == the Python join of those files
        job.join("joined.fa")
        assert open("joined.fa").read() == cli.join_polished(sorted(outs), case["bs"], [c[0] for c in contigs], fast=False)
        # ... and as two processes would write it: one creates, each writes the records it holds
        lens, have = job.polished_lens()
        assert have.all() and [int(v) for v in lens] == [len(t) for t in polished]
        a, b = AssemblyJob.open(path), AssemblyJob.open(path)
        for j in (a, b):
            j.split(case["bs"], "asm.fa", write_files=False)
        for c in range(n_chunks):
            (a if (int(job.chunk_file[c]) % 2 == 0) else b).put(c, polished[c])
        a.join("joined2.fa", all_lens=lens, mode=1)
        b.join("joined2.fa", all_lens=lens, mode=2)
        a.join("joined2.fa", all_lens=lens, mode=2)
        assert open("joined2.fa", "rb").read() == open("joined.fa", "rb").read()
    finally:
        os.chdir(cwd)


def test_only_the_listed_batch_files_are_written(tmp_path):
    rng = random.Random(3)
    path = str(tmp_path / "asm.fa")
    open(path, "w").write(_fasta(rng, [("c%d" % i, 500) for i in range(10)], 60))
    job = AssemblyJob.open(path)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        _, n_files = job.split(400, "q.fa", only_files=[1, 3])
        job.split_wait()
        assert n_files > 4
        assert sorted(p for p in os.listdir(".") if p.startswith("q.fa.batch.")) == ["q.fa.batch.1.fa", "q.fa.batch.3.fa"]
    finally:
        os.chdir(cwd)


@pytest.mark.parametrize("text", [
    ">c1\r\nACGT\r\n",                    # DOS line ends
    ">c1\nAC GT\n",                      # a blank in a sequence line (perl -a takes the first token)
    ">c1\nACGT\tx\n",
    "ACGT\n>c1\nACGT\n",                 # text before the first header
    ">c1\nAC\xc3\xa9GT\n",               # non-ASCII
    ">c1\nACGT\n>c1\nGGGG\n",            # a name twice: the join's hash keeps the last record
    " >c1\nACGT\n",                      # '>' after a blank
    ">c\x0b1 d\nACGT\n",                 # a vertical tab in a header line (str.split() splits there, perl -a does not)
    "",
])
def test_anything_but_the_ordinary_file_is_left_to_the_line_by_line_rules(tmp_path, text):
    path = str(tmp_path / "odd.fa")
    open(path, "wb").write(text.encode("latin-1") if isinstance(text, str) else text)
    assert AssemblyJob.open(path) is None


def test_missing_file_gives_none(tmp_path):
    assert AssemblyJob.open(str(tmp_path / "nosuch.fa")) is None


def test_split_reports_a_directory_that_cannot_be_written(tmp_path):
    rng = random.Random(5)
    path = str(tmp_path / "asm.fa")
    open(path, "w").write(_fasta(rng, [("c", 500)], 60))
    job = AssemblyJob.open(path)
    job.split(100, str(tmp_path / "nosuchdir" / "q.fa"))
    with pytest.raises(RuntimeError):
        job.split_wait()
