#!/usr/bin/env python3
"""Randomised differential test: oracle/jasper_oracle.c  vs  the REAL reference (Jellyfish 2.3.0 + src/jasper.py).

Build-container only (needs /root/reference and the Jellyfish build of SURVEY.md Appendix C). Nothing is
written into the repo; a failing case is left in /tmp/fuzz_fail_<seed> for inspection. This is how the oracle
was pinned beyond the committed golden cases (see DESIGN.md "Oracle").

usage: python3 tests/golden/fuzz_vs_reference.py [--n 200] [--seed0 0]
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402
from oracle import oracle as O  # noqa: E402


def random_case(seed):
    rng = np.random.default_rng(seed)
    # JASPER_FUZZ_WIDE_K=1: keys of more than 86 bits as well (the default list keeps the cases of the recorded seed ranges)
    ks = [15, 17, 20, 21, 24, 25, 28, 31, 33, 36, 37, 41] + ([38, 45, 51, 57, 63] if os.environ.get("JASPER_FUZZ_WIDE_K") else [])
    k = int(rng.choice(ks))
    rl = int(max(k + 10, rng.choice([60, 80, 100, 150])))
    L = int(rng.integers(300, 5000))
    t = G.rand_seq(rng, L)
    # optional structure: homopolymers, tandem repeats, a duplicated segment
    t = list(t)
    for _ in range(int(rng.integers(0, 6))):
        p = int(rng.integers(0, max(1, L - 20)))
        r = int(rng.integers(3, 15))
        t[p:p + r] = ["ACGT"[int(rng.integers(0, 4))]] * r
    t = "".join(t)[:L]
    if rng.random() < 0.3 and L > 800:
        a = int(rng.integers(0, L - 400))
        b = int(rng.integers(0, L - 400))
        seg = t[a:a + int(rng.integers(50, 300))]
        t = t[:b] + seg + t[b + len(seg):]
    haps = [(t, 1.0)]
    if rng.random() < 0.3:
        hp = [(int(p), "sub", 1 + int(rng.integers(0, 3))) for p in rng.choice(L, max(1, L // 300), replace=False)]
        haps = [(t, 0.5), (G.mutate(rng, t, hp), 0.5)]
    if rng.random() < 0.25:  # high-copy element for the rolling threshold
        unit = t[L // 3:L // 3 + min(400, L // 4)]
        v = G.mutate(rng, unit, [(len(unit) // 2, "sub", 1)])
        haps += [(G.rand_seq(rng, rl) + unit + G.rand_seq(rng, rl), float(rng.integers(5, 40)))]
        if rng.random() < 0.5:
            haps += [(G.rand_seq(rng, rl) + v + G.rand_seq(rng, rl), 0.3)]
    if rng.random() < 0.25 and L > 1000:  # coverage gap
        g0 = int(rng.integers(100, L - 300))
        g1 = g0 + int(rng.integers(k, 250))
        haps = [(h[:g0], w) for h, w in haps] + [(h[g1:], w) for h, w in haps]
    nerr = int(rng.integers(0, max(2, L // 60)))
    plan = []
    for p in rng.choice(max(1, L - 2), min(nerr, max(1, L - 2)), replace=False):
        kd = ["sub", "ins", "del", "lower", "set"][int(rng.choice(5, p=[0.45, 0.2, 0.2, 0.05, 0.1]))]
        arg = {"sub": 1 + int(rng.integers(0, 3)), "ins": "ACGT"[int(rng.integers(0, 4))] * int(rng.integers(1, 3)),
               "del": int(rng.integers(1, 3)), "lower": int(rng.integers(1, 80)),
               "set": ["N", "n", "NNNN", "R", "N" * 40, "-", "nN"][int(rng.integers(0, 7))]}[kd]
        plan.append((int(p), kd, arg))
    asm = G.mutate(rng, t, plan)
    # split into chunks like jasper.sh does (fixed size) + sometimes tiny pieces
    chunks = []
    if rng.random() < 0.5:
        bs = int(rng.integers(max(2 * k, 100), max(2 * k + 1, len(asm))))
        for ci in range(0, len(asm), bs):
            chunks.append(("c:%d" % ci, asm[ci:ci + bs]))
    else:
        chunks.append(("c:0", asm))
    if rng.random() < 0.2:
        chunks.append(("tiny:0", asm[:int(rng.integers(0, 2 * k + 3))]))
    spec = dict(k=k, passes=int(rng.integers(1, 4)), thre=int(rng.integers(2, 9)), rl=rl,
                cov=int(rng.integers(15, 60)), err=float(rng.choice([0, 0.001, 0.003, 0.01])),
                fmt=str(rng.choice(["fq", "fa", "fa_multiline", "fq_crlf", "fq_multiline"])))
    return rng, spec, haps, chunks


def run_case(seed, keep=False):
    rng, spec, haps, chunks = random_case(seed)
    k = spec["k"]
    reads = G.sample_reads(rng, haps, spec["cov"], spec["rl"], spec["err"])
    if not reads:
        reads = [haps[0][0][:spec["rl"]]]
    work = tempfile.mkdtemp(prefix="fuzz_")
    ext = "fq" if spec["fmt"].startswith("fq") else "fa"
    rpath = os.path.join(work, "reads." + ext)
    G.write_reads(rpath, reads, spec["fmt"], rng)
    with open(os.path.join(work, "batch.fa"), "w") as f:
        for nm, s in chunks:
            f.write(">%s\n%s\n" % (nm, s))
    env = dict(os.environ, PATH=os.path.dirname(G.JF_BIN) + ":" + os.environ["PATH"])
    db = os.path.join(work, "db.jf")
    G.run([G.JF_BIN, "count", "-C", "-s", "50000", "-m", str(k), "-o", db, "-t", "2", rpath], env=env)
    dump = subprocess.run([G.JF_BIN, "dump", "-c", db], check=True, capture_output=True, env=env).stdout.decode()
    histo = subprocess.run([G.JF_BIN, "histo", db], check=True, capture_output=True, env=env).stdout.decode()
    args = dict(query="batch.fa", k=k, fout="b.fix.csv", ff="b.fixed.fa", db="db.jf", thre=spec["thre"], passes=spec["passes"])
    drv = os.path.join(work, "drv.py")
    open(drv, "w").write(G.DRIVER % dict(jfpy=G.JF_PY, ref=G.REF))
    p = subprocess.run([sys.executable, drv, json.dumps(args)], cwd=work, capture_output=True, text=True)
    # ---- oracle
    odb = O.OracleDB(k)
    odb.count_text(open(rpath, "rb").read())
    problems = []
    ref_counts = {a: int(b) for a, b in (ln.split() for ln in dump.splitlines())}
    if dict(odb.items()) != ref_counts:
        problems.append("counts")
    h = odb.histo()
    if [(m, h[m]) for m in range(1, 10002) if h[m]] != [tuple(int(x) for x in ln.split()) for ln in histo.splitlines()]:
        problems.append("histo")
    names = [c[0] for c in chunks]
    seqs = [c[1] for c in chunks]
    # parse_fasta dict semantics: duplicate names collapse (none generated here)
    try:
        fixed, rows, qv, _ = odb.polish_batch(names, seqs, spec["thre"], spec["passes"])
        oexit = 0
    except RuntimeError:
        oexit = 1
    if (p.returncode != 0) != (oexit != 0):
        problems.append("exit ref=%d oracle=%d" % (p.returncode, oexit))
    elif oexit == 0:
        P = spec["passes"]
        sys.path.insert(0, os.path.join(HERE, ".."))
        from golden_util import CSV_HEADER, fasta60
        if fasta60(names, fixed) != open(os.path.join(work, "_iter%d_b.fixed.fa" % (P - 1))).read():
            problems.append("fixed fasta")
        q0 = [int(x) for x in open(os.path.join(work, "0qValCalcHelper.csv")).read().split()]
        qP = [int(x) for x in open(os.path.join(work, "%dqValCalcHelper.csv" % P)).read().split()]
        if tuple(q0 + qP) != qv:
            problems.append("qv ref=%s oracle=%s" % (q0 + qP, qv))
        for it in range(P):
            if CSV_HEADER + rows[it] != open(os.path.join(work, "_iter%d_b.fix.csv" % it), "rb").read().decode():
                problems.append("csv pass %d" % it)
    if problems and keep:
        dst = "/tmp/fuzz_fail_%d" % seed
        if os.path.isdir(dst):
            shutil.rmtree(dst)
        shutil.copytree(work, dst)
        json.dump(spec, open(os.path.join(dst, "spec.json"), "w"))
    shutil.rmtree(work)
    return problems, spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100)
    ap.add_argument("--seed0", type=int, default=0)
    a = ap.parse_args()
    bad = 0
    for s in range(a.seed0, a.seed0 + a.n):
        pr, spec = run_case(s, keep=True)
        if pr:
            bad += 1
            print("seed", s, "k", spec["k"], "P", spec["passes"], "FAIL", pr, flush=True)
    print("done: %d cases, %d failing" % (a.n, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
