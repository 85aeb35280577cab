"""GPU: the drop-in driver (jasper_amd.cli) against one run of the REAL src/jasper.sh (tests/golden/e2e, produced by
tests/golden/make_golden.py:make_e2e with bash + perl + Jellyfish 2.3.0 + unmodified jasper.py)."""
import gzip
import json
import os
import re
import shutil
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
E2E = os.path.join(HERE, "golden", "e2e")
ROOT = os.path.dirname(HERE)


def fasta_records(path):
    d, name = {}, None
    for ln in open(path):
        if ln.startswith(">"):
            name = ln.strip()
            d[name] = ""
        else:
            d[name] += ln.strip()
    return d


@pytest.mark.parametrize("fixture", ["e2e", "e2e_k45"])
def test_cli_matches_jasper_sh(hip, tmp_path, fixture):
    """e2e: k = 25; e2e_k45: k = 45 -- keys of 90 bits through the whole driver, mer_counts45.jf written and reused"""
    E2E = os.path.join(HERE, "golden", fixture)
    meta = json.load(open(os.path.join(E2E, "meta.json")))
    K = meta["k"]
    for fn in ("r1.fq", "r2.fq"):
        with open(tmp_path / fn, "wb") as f:
            f.write(gzip.open(os.path.join(E2E, fn + ".gz")).read())
    shutil.copy(os.path.join(E2E, "asm.fa"), tmp_path)
    env = dict(os.environ, PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, "-m", "jasper_amd.cli", "-r", "r1.fq r2.fq", "-a", "asm.fa", "-k", str(meta["k"]),
                        "-t", str(meta["threads"]), "-p", str(meta["passes"]), "-d"],
                       cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    # artefacts of src/jasper.sh
    assert open(tmp_path / "threshold.txt").read() == open(os.path.join(E2E, "threshold.txt")).read()
    assert open(tmp_path / ("jfhisto%d.csv" % K)).read() == open(os.path.join(E2E, "jfhisto%d.csv" % K)).read()
    # contig order in the reference is perl-hash order: compare per record
    assert fasta_records(tmp_path / "asm.fa.polished.fasta") == fasta_records(os.path.join(E2E, "asm.fa.polished.fasta"))
    assert open(tmp_path / "asm.fa.fixes.csv", newline="").read() == open(os.path.join(E2E, "asm.fa.fixes.csv"), newline="").read()
    # batch file layout (kept because of -d)
    got = {}
    for fn in sorted(os.listdir(tmp_path)):
        if re.match(r"asm\.fa\.batch\.\d+\.fa$", fn):
            got[fn] = [ln.strip() for ln in open(tmp_path / fn) if ln.startswith(">")]
    assert got == meta["batches"]
    assert sorted(fn for fn in os.listdir(tmp_path) if re.match(r"jasper\..*\.success$", fn)) == meta["sentinels"]
    # log lines: same messages in the same order.  The reference's own Q values read "Inf" (bc is missing where it ran), so the
    # digits are compared with the internal known answers: the oracle's (bad, total) for this fixture through jasper_amd.qv
    mine = [re.sub(r"^\[[^\]]*\]", "[DATE]", ln) for ln in p.stdout.splitlines()]
    strip_q = lambda ls: [re.sub(r"Q value = .*", "Q value =", ln) for ln in ls]
    assert strip_q(mine) == strip_q(meta["stdout"])
    kat = json.load(open(os.path.join(HERE, "golden", "qv_kats.json")))["fixtures"][fixture]
    assert [ln.split("] ", 1)[1] for ln in mine if "Q value" in ln] == ["Before Polishing: Q value = " + kat["before"][2], "After Polishing: Q value = " + kat["after"][2]]
    # src/jasper.sh:177 leaves the database behind (`tee $JF_DB`); ours is a Jellyfish binary/sorted file too
    assert os.path.getsize(tmp_path / ("mer_counts%d.jf" % K)) > 1000
    # a second run in the same directory resumes from the sentinels and leaves the result untouched
    p2 = subprocess.run([sys.executable, "-m", "jasper_amd.cli", "-r", "r1.fq r2.fq", "-a", "asm.fa", "-k", str(K), "-t", "4", "-p", "2"],
                        cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert p2.returncode == 0
    # ... and a run without the sentinels reuses the database (:171-173) and arrives at the same result
    for fn in os.listdir(tmp_path):
        if re.match(r"jasper\..*\.success$", fn) or fn.endswith(".polished.fasta") or fn.endswith(".fixes.csv") or fn.startswith("jfhisto"):
            os.remove(tmp_path / fn)
    p3 = subprocess.run([sys.executable, "-m", "jasper_amd.cli", "-r", "r1.fq r2.fq", "-a", "asm.fa", "-k", str(K), "-t", str(meta["threads"]), "-p", "2"],
                        cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert p3.returncode == 0, p3.stdout + p3.stderr
    assert "Using existing jellyfish database mer_counts%d.jf" % K in p3.stdout
    assert open(tmp_path / ("jfhisto%d.csv" % K)).read() == open(os.path.join(E2E, "jfhisto%d.csv" % K)).read()
    assert open(tmp_path / "asm.fa.fixes.csv", newline="").read() == open(os.path.join(E2E, "asm.fa.fixes.csv"), newline="").read()
    assert fasta_records(tmp_path / "asm.fa.polished.fasta") == fasta_records(os.path.join(E2E, "asm.fa.polished.fasta"))


def test_cli_database_written_before_polishing_when_memory_is_short(hip, tmp_path):
    """src/jasper.sh:177 `tee $JF_DB`: the database file is written by a thread beside the polishing only when the device has room
    for both; otherwise first the file, then the polishing (the reference's order).  Both orders leave the same file and results."""
    E2E = os.path.join(HERE, "golden", "e2e")
    meta = json.load(open(os.path.join(E2E, "meta.json")))
    outs = {}
    for mode in ("beside", "before"):
        d = tmp_path / mode
        d.mkdir()
        for fn in ("r1.fq", "r2.fq"):
            with open(d / fn, "wb") as f:
                f.write(gzip.open(os.path.join(E2E, fn + ".gz")).read())
        shutil.copy(os.path.join(E2E, "asm.fa"), d)
        env = dict(os.environ, PYTHONPATH=ROOT, JASPER_AMD_TIMING="1")
        if mode == "before":
            env["JASPER_AMD_TEST_JF_SERIAL"] = "1"
        p = subprocess.run([sys.executable, "-m", "jasper_amd.cli", "-r", "r1.fq r2.fq", "-a", "asm.fa", "-k", str(meta["k"]), "-t", str(meta["threads"]),
                            "-p", str(meta["passes"])], cwd=d, env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout + p.stderr
        assert ("write mer_counts.jf (before the polishing" in p.stderr) == (mode == "before"), p.stderr
        assert ("mer_counts.jf complete (written beside" in p.stderr) == (mode == "beside"), p.stderr
        raw = open(d / ("mer_counts%d.jf" % meta["k"]), "rb").read()
        hlen = int(raw[:9])
        outs[mode] = (raw[9 + hlen:], open(d / "asm.fa.fixes.csv", newline="").read(), fasta_records(d / "asm.fa.polished.fasta"))
        assert not os.path.exists(d / ("mer_counts%d.jf.tmp" % meta["k"]))
    assert outs["beside"] == outs["before"]
    assert outs["before"][2] == fasta_records(os.path.join(E2E, "asm.fa.polished.fasta"))


def test_cli_errors(hip, tmp_path):
    env = dict(os.environ, PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, "-m", "jasper_amd.cli", "-a", "missing.fa"], cwd=tmp_path, env=env, capture_output=True, text=True)
    assert p.returncode == 1 and "The query file does not exist" in p.stderr
    (tmp_path / "a.fa").write_text(">c\nACGT\n")
    p = subprocess.run([sys.executable, "-m", "jasper_amd.cli", "-a", "a.fa", "-r", "nope.fq"], cwd=tmp_path, env=env, capture_output=True, text=True)
    assert p.returncode == 1 and "does not exist" in p.stderr


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torchrun_cli(cwd, args, nproc=2, extra_env=None):
    """the multi-GPU form of the driver: one process per GPU under torch.distributed.run; rehearsed here with both ranks on
    the one GPU of the test box and gloo as the transport (RCCL refuses two ranks on one device)"""
    env = dict(os.environ, PYTHONPATH=ROOT, JASPER_AMD_DIST_BACKEND="gloo", JASPER_AMD_ONE_GPU="1", JASPER_AMD_TIMING="1")
    env.update(extra_env or {})
    if nproc > 2:
        # the test box allows 6 processes on its GPU: pytest + the launcher + nproc ranks leave no room for the throw-away
        # process that probes the IPC mapping (dist._ipc_probe_ok); two-rank runs keep it
        env["JASPER_AMD_IPC_PROBE"] = "0"
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
                           "--master-port", str(_free_port()), "-m", "jasper_amd.cli"] + args,
                          cwd=cwd, env=env, capture_output=True, text=True, timeout=240)


def test_cli_two_ranks_matches_jasper_sh(hip, tmp_path):
    """read files cut into per-rank byte ranges, counts summed by key owner, batch files divided over the ranks, lookups
    through IPC-mapped owner tables: same artefacts as the real jasper.sh run"""
    meta = json.load(open(os.path.join(E2E, "meta.json")))
    for fn in ("r1.fq", "r2.fq"):
        with open(tmp_path / fn, "wb") as f:
            f.write(gzip.open(os.path.join(E2E, fn + ".gz")).read())
    shutil.copy(os.path.join(E2E, "asm.fa"), tmp_path)
    args = ["-r", "r1.fq r2.fq", "-a", "asm.fa", "-k", str(meta["k"]), "-t", str(meta["threads"]), "-p", str(meta["passes"]), "-d"]
    p = _torchrun_cli(tmp_path, args, extra_env={"JASPER_AMD_COUNT": "exchange"})
    assert p.returncode == 0, p.stdout + p.stderr
    # counted without a table per GPU: file reader -> batches of bases -> region lists by key owner -> one all_to_all -> owners' shards
    assert "region lists -> owners' shards" in p.stderr and "local table" not in p.stderr, p.stderr

    def check_outputs():
        assert open(tmp_path / "threshold.txt").read() == open(os.path.join(E2E, "threshold.txt")).read()
        assert open(tmp_path / "jfhisto25.csv").read() == open(os.path.join(E2E, "jfhisto25.csv")).read()
        assert fasta_records(tmp_path / "asm.fa.polished.fasta") == fasta_records(os.path.join(E2E, "asm.fa.polished.fasta"))
        assert open(tmp_path / "asm.fa.fixes.csv", newline="").read() == open(os.path.join(E2E, "asm.fa.fixes.csv"), newline="").read()
    check_outputs()
    assert sorted(fn for fn in os.listdir(tmp_path) if re.match(r"jasper\..*\.success$", fn)) == meta["sentinels"]
    mine = [re.sub(r"^\[[^\]]*\]", "[DATE]", ln) for ln in p.stdout.splitlines() if re.match(r"^\[\w{3} \w{3} +\d", ln)]   # (gloo prints its own "[Gloo] ..." lines)
    strip_q = lambda ls: [re.sub(r"Q value = .*", "Q value =", ln) for ln in ls]
    assert strip_q(mine) == strip_q(meta["stdout"])                       # only rank 0 talks, same lines as one process
    kat = json.load(open(os.path.join(HERE, "golden", "qv_kats.json")))["fixtures"]["e2e"]      # the Q digits: internal known answers
    assert [ln.split("] ", 1)[1] for ln in mine if "Q value" in ln] == ["Before Polishing: Q value = " + kat["before"][2], "After Polishing: Q value = " + kat["after"][2]]
    # src/jasper.sh:177 leaves the database behind: written by both ranks together, one consecutive sorted piece each.
    # The reader's ordering rule (test_gpu_jf.py restates it from binary_dumper.hpp) holds across the seam, the content is
    # what one GPU writes, and the file goes back to the build container for the real jellyfish to read (gpurun_out)
    sys.path.insert(0, HERE)
    from test_gpu_jf import _read_jf, _jf_pos
    from jasper_amd import KmerTable
    db2 = tmp_path / "mer_counts25.jf"
    hdr2, recs2 = _read_jf(str(db2))
    order = [(_jf_pos(hdr2, k), k) for k, _ in recs2]
    assert order == sorted(order) and len(set(order)) == len(order) and hdr2["size"] == 1 << hdr2["matrix1"]["r"]
    assert hdr2["cmdline"][:2] == ["count", "-C"] and hdr2["cmdline"][-2:] == ["r1.fq", "r2.fq"]
    t2 = KmerTable.from_jf(str(db2))
    assert ["%d %d" % (m, n) for m, n in t2.histo_rows()] == open(os.path.join(E2E, "jfhisto25.csv")).read().split("\n")[:-1]
    t2.close()
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        shutil.copy(db2, os.path.join(out, "two_rank_mer_counts25.jf"))
    os.remove(db2)

    # an existing database is read in record ranges, one per rank: write it with a one-GPU run, wipe, run two ranks again
    def wipe():
        for fn in os.listdir(tmp_path):
            if re.match(r"jasper\..*\.success$", fn) or fn.endswith(".polished.fasta") or fn.endswith(".fixes.csv") or fn.startswith("jfhisto") \
                    or fn == "threshold.txt" or ".batch." in fn:
                os.remove(tmp_path / fn)
    wipe()
    env = dict(os.environ, PYTHONPATH=ROOT)
    p1 = subprocess.run([sys.executable, "-m", "jasper_amd.cli"] + args, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert p1.returncode == 0 and os.path.getsize(tmp_path / "mer_counts25.jf") > 1000
    wipe()
    p2 = _torchrun_cli(tmp_path, args)
    assert p2.returncode == 0, p2.stdout + p2.stderr
    assert "Using existing jellyfish database mer_counts25.jf" in p2.stdout
    check_outputs()
    # ... and the same through -j, started with the driver's own --gpus flag
    wipe()
    os.rename(tmp_path / "mer_counts25.jf", tmp_path / "db.jf")
    env2 = dict(os.environ, PYTHONPATH=ROOT, JASPER_AMD_DIST_BACKEND="gloo", JASPER_AMD_ONE_GPU="1", MASTER_PORT=str(_free_port()))
    p3 = subprocess.run([sys.executable, "-m", "jasper_amd.cli", "-j", "db.jf", "-a", "asm.fa", "-k", str(meta["k"]), "-t", str(meta["threads"]),
                         "-p", str(meta["passes"]), "--gpus", "2"], cwd=tmp_path, env=env2, capture_output=True, text=True, timeout=900)
    assert p3.returncode == 0, p3.stdout + p3.stderr
    check_outputs()


@pytest.mark.parametrize("nproc,count", [(3, "exchange"), (3, "local")] + ([(4, "exchange")] if os.environ.get("JASPER_TEST_BIG") else []))    # (4 ranks sit at the test box's process limit)
def test_cli_three_ranks(hip, tmp_path, nproc, count):
    """rank counts beyond two, one of them not a power of two: that many byte ranges per read file, key owners, file pieces;
    counted by the exchange of region lists, and (JASPER_AMD_COUNT=local) into a table per GPU whose entries are then summed
    by owner"""
    meta = json.load(open(os.path.join(E2E, "meta.json")))
    for fn in ("r1.fq", "r2.fq"):
        with open(tmp_path / fn, "wb") as f:
            f.write(gzip.open(os.path.join(E2E, fn + ".gz")).read())
    shutil.copy(os.path.join(E2E, "asm.fa"), tmp_path)
    p = _torchrun_cli(tmp_path, ["-r", "r1.fq r2.fq", "-a", "asm.fa", "-k", str(meta["k"]), "-t", str(meta["threads"]), "-p", str(meta["passes"])], nproc=nproc,
                      extra_env={"JASPER_AMD_COUNT": count})
    assert p.returncode == 0, p.stdout + p.stderr
    assert ("local table" in p.stderr) == (count == "local"), p.stderr
    assert open(tmp_path / "threshold.txt").read() == open(os.path.join(E2E, "threshold.txt")).read()
    assert open(tmp_path / "jfhisto25.csv").read() == open(os.path.join(E2E, "jfhisto25.csv")).read()
    assert fasta_records(tmp_path / "asm.fa.polished.fasta") == fasta_records(os.path.join(E2E, "asm.fa.polished.fasta"))
    assert open(tmp_path / "asm.fa.fixes.csv", newline="").read() == open(os.path.join(E2E, "asm.fa.fixes.csv"), newline="").read()
    sys.path.insert(0, HERE)
    from test_gpu_jf import _read_jf, _jf_pos
    hdr, recs = _read_jf(str(tmp_path / "mer_counts25.jf"))
    order = [(_jf_pos(hdr, k), k) for k, _ in recs]
    assert order == sorted(order) and len(set(order)) == len(order)


@pytest.mark.parametrize("count", ["local", "exchange"])
def test_cli_two_ranks_degenerate_inputs(hip, tmp_path, count):
    """a rank with nothing to count, a rank with no batch file, and the run the reference aborts (no usable threshold):
    two ranks behave like one process, whichever way the counts reach the key owners"""
    rng = __import__("numpy").random.default_rng(4)
    genome = "".join(rng.choice(list("ACGT"), 3000))
    (tmp_path / "asm.fa").write_text(">c1\n" + genome + "\n")
    # one read only: rank 1's byte range of the file is empty; the histogram has a single row -> threshold script exits 1
    (tmp_path / "one.fq").write_text("@r\n%s\n+\n%s\n" % (genome[100:250], "I" * 150))
    args = ["-r", "one.fq", "-a", "asm.fa", "-k", "25", "-t", "1", "-p", "1"]
    env = dict(os.environ, PYTHONPATH=ROOT)
    one = tmp_path / "single"
    two = tmp_path / "double"
    for d in (one, two):
        d.mkdir()
        for fn in ("asm.fa", "one.fq"):
            shutil.copy(tmp_path / fn, d)
    p1 = subprocess.run([sys.executable, "-m", "jasper_amd.cli"] + args, cwd=one, env=env, capture_output=True, text=True, timeout=300)
    p2 = _torchrun_cli(two, args, extra_env={"JASPER_AMD_COUNT": count})
    assert p1.returncode == 1 and "Local min of kmer counts is smaller than 4" in p1.stderr
    assert p2.returncode != 0 and "Local min of kmer counts is smaller than 4" in p2.stderr
    assert open(one / "jfhisto25.csv").read() == open(two / "jfhisto25.csv").read() == "1 126\n"
    # the golden reads (usable threshold) with an assembly that makes ONE batch file: rank 1 has nothing to polish
    first = fasta_records(os.path.join(E2E, "asm.fa"))
    name, seq = sorted(first.items())[0]
    for d in (one, two):
        for fn in os.listdir(d):
            os.remove(d / fn)
        # (one plain file that is cut in two byte ranges, and one gzip file, which goes whole to one rank)
        with open(d / "many.fq", "wb") as f:
            f.write(gzip.open(os.path.join(E2E, "r1.fq.gz")).read())
        shutil.copy(os.path.join(E2E, "r2.fq.gz"), d / "r2.fq.gz")
        (d / "asm.fa").write_text("%s\n%s\n" % (name, seq[:3000]))
    args = ["-r", "many.fq r2.fq.gz", "-a", "asm.fa", "-k", "25", "-t", "1", "-p", "2"]
    p1 = subprocess.run([sys.executable, "-m", "jasper_amd.cli"] + args, cwd=one, env=env, capture_output=True, text=True, timeout=300)
    p2 = _torchrun_cli(two, args, extra_env={"JASPER_AMD_COUNT": count})
    assert p1.returncode == 0 and p2.returncode == 0, p1.stderr + p2.stdout + p2.stderr
    for fn in ("asm.fa.polished.fasta", "asm.fa.fixes.csv", "threshold.txt", "jfhisto25.csv"):
        assert open(one / fn, newline="").read() == open(two / fn, newline="").read(), fn
    assert len([fn for fn in os.listdir(two) if fn.startswith("jasper.") and fn.endswith(".success")]) == 5
