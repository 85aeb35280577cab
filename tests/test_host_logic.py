"""CPU: host-side mirrors of src/jasper.sh / src/jasper.py / src/jellyfish.py logic (no GPU, no library calls)."""
import os

import pytest

from golden_util import Case, case_names
from jasper_amd import cli, polisher, qv, synth, dist
from oracle import oracle as O


@pytest.mark.parametrize("name", case_names())
def test_threshold_mirror_matches_reference_script(name):
    c = Case(name)
    txt, status = polisher.threshold_from_histo_rows(c.histo_rows())
    assert status == c.meta["jellyfish_py_exit"] and txt == c.meta["jellyfish_py_stdout"]


def test_threshold_edge_cases():
    f = polisher.threshold_from_histo_rows
    assert f([]) == ("", 0)
    assert f([(1, 10)]) == ("", 0)
    assert f([(1, 100), (2, 50), (3, 60)]) == ("", 1)             # local min at 2 -> int(2/2)=1 < 2 -> exit 1
    assert f([(1, 100), (2, 50), (3, 20), (4, 10), (5, 12)]) == ("2", 0)
    assert f([(1, 100), (2, 50), (3, 20), (4, 10), (5, 9), (6, 30)]) == ("2", 0)
    assert f([(1, 5), (2, 9)]) == ("", 1)                           # first row only sets count; rise at row 2 with threshold 0
    assert f([(1, 100), (2, 50), (3, 20)]) == ("", 0)              # never rises: prints nothing, exit 0
    for rows in ([(1, 100), (2, 50), (3, 60)], [(1, 100), (2, 50), (3, 20), (4, 10), (5, 12)], [(1, 5), (2, 9)]):
        try:
            t = O.threshold(rows)
            o = (str(t) if t else "", 0)
        except SystemExit:
            o = ("", 1)
        assert o == f(rows)


def test_step_rule():
    """src/jasper.py:20 step = max(2, round(k/8)) with python's round-half-to-even"""
    import math
    for k in range(6, 64):
        assert max(2, round(k / 8)) == max(2, int(math.floor(k / 8 + 0.5)) if (k / 8) % 1 != 0.5 else max(2, round(k / 8)))


def test_batch_size_rule():
    assert synth.jasper_batch_size(1_000_000, 4) == 225000            # SURVEY A.5 probe
    assert synth.jasper_batch_size(47_000_000, 16) == 2643750
    assert synth.jasper_batch_size(3_100_000_000, 16) == 25000000     # capped
    assert synth.jasper_batch_size(1000, 2, user_batch=30000000) == 30000000   # a larger user -b is not capped


def test_split_and_join_roundtrip(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    contigs = [(">c0", "A" * 600000), (">c1", "C" * 300000), (">c2", "G" * 100000)]
    files = cli.split_batches(contigs, 225000, "asm.fa")
    got = [[ln.strip() for ln in open(f) if ln.startswith(">")] for f in files]
    # SURVEY A.5 probe: {c0:0,c0:225000}, {c0:450000,c1:0}, {c1:225000,c2:0}
    assert got == [[">c0:0", ">c0:225000"], [">c0:450000", ">c1:0"], [">c1:225000", ">c2:0"]]
    text = cli.join_polished(files, 225000, [c[0] for c in contigs])
    assert text == "".join("%s\n%s\n" % c for c in contigs)


def test_what_a_stage_remembers_of_its_own_files_is_what_reading_them_gives(tmp_path, monkeypatch):
    """the batch files are written and then polished by the same process: their records are kept (polisher.remember) instead of
    parsed again -- the same dict parse_fasta makes, for as long as the file is what was written; duplicate chunk names (two contigs
    whose headers share the first token) included"""
    import os
    import time
    monkeypatch.chdir(tmp_path)
    contigs = [(">c0", "ACGT" * 1000), (">c0", "TTTT" * 300), (">c1", "G" * 2500), (">odd:name", "N" * 10)]
    files = cli.split_batches(contigs, 1500, "asm.fa")
    assert len(files) > 2
    for f in files:
        kept = polisher.recall(f, "records")
        assert kept is not None and list(kept.items()) == list(polisher.parse_fasta(f).items())
        assert polisher.recall(f, "events") is None
    # a file that changed after it was written is read again
    time.sleep(0.01)
    with open(files[0], "a") as f:
        f.write(">late\nAC\n")
    assert polisher.recall(files[0], "records") is None
    # 60-column wrapping through numpy == the reference's slicing (src/jasper.py:142-147)
    for s in ["", "A", "ACGT" * 15, "ACGT" * 15 + "A", "ACGTN" * 1000]:
        lines = polisher.split_output(s, 60)
        assert polisher.wrap_lines(s, 60) == (("\n".join(lines) + "\n").encode() if lines else b"")
    assert polisher.wrap_lines("ACG\u00e9T") is None


def test_read_assembly_first_token_rules(tmp_path):
    p = tmp_path / "a.fa"
    p.write_text(">c0 desc here\nACGT extra\nTTTT\n\n>c1\nGG\n>empty\n>c2\nA\n")
    assert cli.read_assembly(str(p)) == [(">c0", "ACGTTTTT"), (">c1", "GG"), (">c2", "A")]
    assert cli.sequence_bytes(str(p)) == len("ACGT extra") + 4 + 0 + 2 + 1


def test_parse_fasta_dict_semantics(tmp_path):
    p = tmp_path / "b.fa"
    p.write_text("junk before\n>a x y\nAC\nGT\n>b\n\n>a\nTT\n")
    d = polisher.parse_fasta(str(p))
    assert list(d.items()) == [("a", "TT"), ("b", "")]


def test_rows_and_csv_format():
    rows = []
    rows += polisher.rows_from_record("c:0", dict(kind="s", index=12, newc="A", oldc="c", rep=1))
    rows += polisher.rows_from_record("c:0", dict(kind="i", index=13, newc="-", oldc="T", rep=3))
    rows += polisher.rows_from_record("c:0", dict(kind="d", index=14, newc="G", oldc="-", rep=2))
    txt = polisher.fix_csv_text(rows)
    assert txt == "Contig Base_coord Original Mutation\r\nc:0 12 A sc\r\nc:0 13 - iTTT\r\nc:0 14 GG d-\r\n"
    one = polisher.rows_from_record("n", dict(kind="x", index=100, patch="ACGTA", orig="ACCTA"))
    assert polisher.fix_csv_text(one).splitlines()[1] == "n 102 ['G'] ['sC']"
    two = polisher.rows_from_record("n", dict(kind="x", index=5, patch="ACGTA", orig="ACTAT"))
    assert len(two) == 2
    with pytest.raises(IndexError):
        polisher.rows_from_record("n", dict(kind="x", index=5, patch="ACGT", orig="ACGT"))


def test_alignment_standin_properties():
    import random
    rnd = random.Random(3)
    for _ in range(200):
        a = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(0, 30)))
        b = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(0, 30)))
        ra, rb = polisher.globalms_first(a, b)
        assert len(ra) == len(rb)
        assert ra.replace("-", "") == a and rb.replace("-", "") == b
        assert all(not (x == "-" and y == "-") for x, y in zip(ra, rb))


def test_qv_formatting():
    assert qv.q_value(0, 1000, 37) == "Inf"
    s = qv.q_value(171811, 46999330, 37)
    assert s.count(".") == 1 and len(s.split(".")[1]) == 5
    import math
    approx = -10 * math.log10(1 - (1 - 171811 / 46999330) ** (1 / 37))
    assert abs(float(s) - approx) < 1e-3
    assert abs(float(qv.q_value(112, 46999349, 37)) - (-10 * math.log10(1 - (1 - 112 / 46999349) ** (1 / 37)))) < 1e-2


def test_bc_emulation_known_answers():
    """jasper_amd/qv.py restates GNU bc's number arithmetic and libmath's l() / e() (src/jasper.sh:239-256 runs them at scale
    10 / 50 / 5): what every GNU bc prints for a few expressions, and the internal known-answer table of Q strings"""
    import json
    kats = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "qv_kats.json")))
    bc = qv._Bc(20)
    got = {"scale=20; l(2)": bc.l((2, 0)), "scale=20; e(1)": bc.e((1, 0)), "scale=20; sqrt(2)": bc.sqrt((2, 0)), "scale=20; l(10)": bc.l((10, 0))}
    for expr, want in kats["bc_constants"].items():
        assert qv._Bc.show(got[expr]) == want, expr
    assert bc.scale == 20                                   # l() and e() restore the caller's scale
    # the scale rules of * and / (bc manual): 2 decimals x 3 decimals at scale 4 -> 4 decimals, truncated
    b4 = qv._Bc(4)
    assert qv._Bc.show(b4.mul(qv._Bc.lit("1.25"), qv._Bc.lit("-.333"))) == "-.4162"
    assert qv._Bc.show(b4.div((2, 0), (3, 0))) == ".6666" and qv._Bc.show(b4.div((-2, 0), (3, 0))) == "-.6666"
    assert qv._Bc.show(qv._Bc.add(qv._Bc.lit("1.5"), qv._Bc.lit(".25"))) == "1.75"
    for bad, total, k, want in kats["triples"]:
        assert qv.q_value(bad, total, k) == want, (bad, total, k)
    for fx in kats["fixtures"].values():
        for bad, total, want in (fx["before"], fx["after"]):
            assert qv.q_value(bad, total, fx["k"]) == want


def test_bc_emulation_against_the_independent_restatement():
    """the series-based emulation against correctly rounded ln / exp truncated where bc truncates: the two may differ in the
    LAST printed decimal only (-10 x a logarithm cut to 5 decimals, divided by l(10) = 2.30258: at most a few units of 1e-5);
    tools/qv_compare.py runs 10^5 triples (docs/experiments.md has the counts)"""
    import random
    rng = random.Random(7)
    n = differ = 0
    for _ in range(1500):
        total = rng.randint(1000, 4 * 10 ** 9)
        bad = int(total * 10 ** rng.uniform(-7.5, -0.3))
        k = rng.choice([17, 21, 25, 31, 37, 45, 63])
        a, b = qv.q_value(bad, total, k), qv.q_value_exact(bad, total, k)
        n += 1
        if a != b:
            differ += 1
            assert a != "Inf" and b != "Inf" and abs(float(a) - float(b)) < 6e-5, (bad, total, k, a, b)
    assert differ <= n // 100


def test_cli_parser_quirks():
    o = cli.parse_args(["-a", "x/asm.fa", "-k", "25", "-p", "1", "-t", "8", "-b", "123"])
    assert (o.query_fn, o.kmer, o.passes, o.num_threads, o.batch_size) == ("asm.fa", "25", "1", "8", "123")
    o = cli.parse_args(["-k", "25", "-d"])
    assert o.debug and o.kmer == "25"
    with pytest.raises(SystemExit):               # src/jasper.sh:93-96: -d eats the next argument, so "25" is unknown
        cli.parse_args(["-d", "-k", "25"])
    with pytest.raises(SystemExit):
        cli.parse_args(["--nope"])


def test_chunk_assignment_and_shards():
    owner = dist.assign_chunks([10, 9, 8, 1, 1, 1], 2)
    loads = [sum(l for l, o in zip([10, 9, 8, 1, 1, 1], owner) if o == r) for r in range(2)]
    assert sorted(loads) == [13, 17] and sum(loads) == 30     # longest-first greedy
    cover = []
    for r in range(3):
        lo, hi = dist.shard_range(10, r, 3)
        cover += list(range(lo, hi))
    assert cover == list(range(10))


def test_synth_recipe_small():
    import numpy as np
    rng = np.random.default_rng(1)
    g = synth.make_genome(rng, 50000)
    r = synth.make_reads_stream(rng, g, 5, 100, 0.01)
    assert r.size == (50000 * 5 // 100) * 101 and set(np.unique(r)) <= set(b"ACGTN")
    a = synth.make_assembly(rng, g, err=1e-3, n_every=20000, n_len=50)
    assert abs(len(a) - len(g)) < 50 and b"N" * 50 in a.tobytes()


def test_fast_text_paths_equal_the_line_by_line_rules(tmp_path):
    """read_assembly / sequence_bytes / join_polished take whole bodies at once when a file is ordinary (printable ASCII
    sequence lines) and fall back to the perl one-liners' line-by-line rules otherwise: same results on files that are
    anything but ordinary"""
    import random
    from jasper_amd import cli
    rnd = random.Random(7)
    pieces = ["ACGT" * 3, "acgtn", "NNNN", "", "A C", "\tACG", "ACG\t", " >x y", ">c1 desc more", ">c2", ">", "> lead", ">c3:0", ">c3:40", "AC\rGT",
              "ACGT\r", "\x0bAC", "AC\x1cGT", "caf\xc3\xa9", "\xff\xfe", ">n\xc3\xa4me z", "!#%~", ">c4\r", ">a\rb c"]
    files = []
    for i in range(400):
        n = rnd.randrange(0, 12)
        ordinary = rnd.random() < 0.5
        pool = [p for p in pieces if not ordinary or (p.isascii() and not any(ch in p for ch in " \t\r\x0b\x1c") or p.startswith(">") and "\r" not in p)]
        lines = [rnd.choice(pool) for _ in range(n)]
        data = "\n".join(lines).encode("latin-1") + (b"\n" if rnd.random() < 0.7 else b"")
        p = tmp_path / ("f%d.fa" % i)
        p.write_bytes(data)
        files.append(str(p))
        assert cli._fasta_events(str(p), True) == cli._fasta_events(str(p), False), data
        assert cli.read_assembly(str(p), True) == cli.read_assembly(str(p), False), data
        assert cli.sequence_bytes(str(p), True) == cli.sequence_bytes(str(p), False), data
    for j in range(0, 400, 7):
        group = files[j:j + 7]
        order = [">c1", ">c3", ">c2"]
        assert cli.join_polished(group, 40, order, True) == cli.join_polished(group, 40, order, False)
    # an ordinary file does take the fast path (the same events, and no '\r' / blanks in it)
    p = tmp_path / "plain.fa"
    p.write_bytes(b">chr1 some text\n" + b"\n".join([b"ACGT" * 15] * 1000) + b"\n>chr2\nAC\n\nGT")
    ev = cli._fasta_events(str(p))
    assert ev == [("h", ">chr1"), ("s", "ACGT" * 15000), ("h", ">chr2"), ("s", "ACGT")] == cli._fasta_events(str(p), False)
    assert cli.sequence_bytes(str(p)) == 60000 + 4


def test_front_process_passes_the_workers_status_and_output_on(tmp_path):
    """cli._front_process (opt-in, JASPER_AMD_FRONT=1): the command a user waits for is a front that forks the worker and ends on the worker's word; a worker
    that exits with a status (here: the reference's own error exits, src/jasper.sh:35-39,125-128) ends without that word and its
    status and messages are the command's -- the same with and without the front."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for env_extra in ({"JASPER_AMD_FRONT": "1"}, {}):
        env = dict(os.environ, PYTHONPATH=root, **env_extra)
        p = subprocess.run([sys.executable, "-m", "jasper_amd.cli", "-a", "nosuch.fa", "-r", "x.fq"], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=120)
        assert p.returncode == 1
        outs.append(p.stderr.split("] ", 1)[-1])
        h = subprocess.run([sys.executable, "-m", "jasper_amd.cli", "-h"], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=120)
        assert h.returncode == 0 and "Usage: jasper.sh [options]" in h.stdout
    assert outs[0] == outs[1] and "The query file does not exist" in outs[0]


def test_front_process_hands_signals_to_the_worker(tmp_path):
    import subprocess, sys, os, signal, time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = ("import os, sys, time\n"
            "from jasper_amd import cli\n"
            "fd = cli._front_process()\n"
            "assert fd is not None\n"                      # (this is the worker)
            "open('worker.pid', 'w').write(str(os.getpid()))\n"
            "time.sleep(60)\n")
    p = subprocess.Popen([sys.executable, "-c", prog], cwd=tmp_path, env=dict(os.environ, PYTHONPATH=root, JASPER_AMD_FRONT="1"))
    for _ in range(200):
        if os.path.exists(tmp_path / "worker.pid") and open(tmp_path / "worker.pid").read():
            break
        time.sleep(0.05)
    worker = int(open(tmp_path / "worker.pid").read())
    assert worker != p.pid
    p.send_signal(signal.SIGTERM)                         # to the front: it passes the signal on and ends the way the worker ended
    rc = p.wait(timeout=30)
    assert rc in (-signal.SIGTERM, 128 + signal.SIGTERM)
    for _ in range(100):
        try:
            os.kill(worker, 0)
        except ProcessLookupError:
            break
        time.sleep(0.05)
    else:
        os.kill(worker, signal.SIGKILL)
        raise AssertionError("the worker outlived the signal")


def test_merge_fix_csvs_fast_path_equals_the_awk_rules(tmp_path):
    """cli.merge_fix_csvs (src/jasper.sh:222-226): rows of plain ASCII are split with the built-in; every other row takes awk's
    field rule ('\\r', '\\v', '\\f' and non-ASCII blanks are NOT separators) -- both orders and outputs must agree with the rule
    applied to every row."""
    import random, re
    from jasper_amd import cli
    random.seed(11)

    def by_the_rule(files):
        lines = []
        for n, path in enumerate(files):
            content = open(path, "r", newline="").read().split("\n")
            if content and content[-1] == "":
                content.pop()
            lines += [ln for fnr, ln in enumerate(content, start=1) if (n == 0 and fnr == 1) or fnr > 1]
        rows = [(ln.split(":") + [""])[0] + " " + (ln.split(":") + [""])[1] for ln in lines]
        awk = lambda rec: re.split(r"[ \t\n]+", rec.strip(" \t\n")) if rec.strip(" \t\n") else []
        num = lambda x: int(re.match(r"[ \t]*-?\d+", x).group(0)) if re.match(r"[ \t]*-?\d+", x) else 0
        rows.sort(key=lambda s: ((awk(s) + [""])[0].encode(), num(awk(s)[1]) if len(awk(s)) > 1 else 0, num(awk(s)[2]) if len(awk(s)) > 2 else 0, s.encode()))
        return "".join("%s:%s %s %s %s\n" % tuple((awk(s) + [""] * 5)[:5]) for s in rows)

    field = lambda: random.choice(["12", "007", "-3", "x9", "", " 5", "\t7", "9\r", "33 44"] + ([] if trial % 2 else ["a\x0bb", "é1", "²"]))
    native_taken = 0
    for trial in range(120):
        files = []
        for n in range(random.randint(1, 3)):
            path = str(tmp_path / ("f%d_%d.csv" % (trial, n)))
            with open(path, "w", newline="") as f:
                f.write("Contig Base_coord Original Mutated\r\n")
                for _ in range(random.randint(0, 10)):
                    f.write("%s:%s %s %s %s%s" % (random.choice(["ctg1", "ctg2", "c t", "x\ty", "a"]), field(), field(), random.choice("ACGT-"),
                                                  random.choice(["sA", "i-", "d-"]), random.choice(["\r\n", "\n"])))
            files.append(path)
        assert cli.merge_fix_csvs(files) == by_the_rule(files), trial
        # the native merge (jasper_merge_fix_csvs): the same bytes whenever it takes the input (printable ASCII, blanks, tabs, '\r')
        import ctypes as C
        from jasper_amd import _lib
        arr = (C.c_char_p * len(files))(*[p.encode() for p in files])
        out = str(tmp_path / ("merged%d.csv" % trial))
        rc = _lib.lib().jasper_merge_fix_csvs(arr, len(files), out.encode())
        raw = b"".join(open(p, "rb").read() for p in files)
        plain = all((0x20 <= c <= 0x7e) or c in (9, 10, 13) for c in raw)
        assert rc == (0 if plain else 1), (trial, rc)
        if rc == 0:
            native_taken += 1
            assert open(out, "r", newline="").read() == by_the_rule(files), trial
    assert native_taken >= 10


def test_rows_straight_from_the_record_array_equal_rows_from_record():
    """polisher._rows_by_pass_and_chunk (the plain kinds from the array's columns) against rows_from_record on decoded records"""
    import numpy as np
    from jasper_amd import polisher
    from jasper_amd.table import FIXREC_DTYPE, PolishResult
    rng = np.random.default_rng(3)
    n = 500
    raw = np.zeros(n, dtype=FIXREC_DTYPE)
    raw["chunk"] = np.sort(rng.integers(0, 7, n))
    raw["pass_"] = rng.integers(0, 3, n)
    raw["seqno"] = rng.permutation(n)
    raw["index"] = rng.integers(0, 10**7, n)
    raw["kind"] = np.frombuffer(b"sidx", dtype=np.uint8)[rng.choice(4, n, p=[.6, .2, .15, .05])]
    raw["newc"] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)]
    raw["oldc"] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)]
    raw["rep"] = rng.integers(1, 4, n)
    aux = [b""] * 7
    for i in np.flatnonzero(raw["kind"] == ord("x")):
        c = int(raw["chunk"][i])
        orig = "".join(rng.choice(list("ACGT"), 30))
        patch = orig[:10] + "T" + orig[10:17] + orig[19:]            # two differences at least
        raw["aux_off"][i], raw["aux_len"][i], raw["rep"][i] = len(aux[c]), len(patch), len(orig)
        aux[c] += (patch + orig).encode()
    res = PolishResult(None, None, 7, False, raw, aux, (0, 0, 0, 0), 0, 0.0)
    got = polisher._rows_by_pass_and_chunk(res, lambda c: "ctg%d:0" % c)
    want = {}
    for r in sorted(res.records, key=lambda r: (r["chunk"], r["pass_"], r["seqno"])):
        want.setdefault((r["pass_"], r["chunk"]), []).extend(polisher.rows_from_record("ctg%d:0" % r["chunk"], r))
    assert got == want and sum(len(v) for v in got.values()) >= n
