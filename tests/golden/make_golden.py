#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/cases/ by RUNNING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference). What it executes, unmodified:
  * `jellyfish count -C`, `jellyfish histo`, `jellyfish dump -c`  -- Jellyfish 2.3.0 built from
    /root/reference/jellyfish-2.3.0.tar.gz (binary expected in $JF_BIN or on PATH,
    python binding `dna_jellyfish` expected on $JF_PY or PYTHONPATH)
  * /root/reference/src/jellyfish.py  (threshold script)
  * /root/reference/src/jasper.py     (per-batch polisher), imported with importlib and driven through main()

The only thing that is NOT the reference is a stand-in for `Bio.pairwise2` (Biopython is a third-party
dependency that is neither vendored nor installed; single call site src/jasper.py:309). It influences
ONLY which rows the ">k bad k-mers" branch writes to the fix CSV -- never the fixed sequence or the QV
counters -- so those rows are marked unpinned: every case stores `unpinned_rows.json`, the (pass, row)
positions of CSV rows that came out of that branch.

Fixtures are DATA (inputs + reference outputs). No reference source is written anywhere.

usage: python3 tests/golden/make_golden.py [--out tests/golden/cases] [--only NAME | e2e | kats]
       (mer_kats.json and e2e/ are written next to the --out directory)
"""
import argparse
import gzip
import importlib.util
import json
import os
import shutil
import subprocess
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference/src"
JF_BIN = os.environ.get("JF_BIN", "/tmp/jf_install/bin/jellyfish")
JF_PY = os.environ.get("JF_PY", "/tmp/jf_pyall")

# ----------------------------------------------------------------------------------------------
# driver executed in a subprocess: imports the reference's jasper.py and calls its main()
# ----------------------------------------------------------------------------------------------
DRIVER = r'''
import sys, types, importlib.util, json
sys.path.insert(0, %(jfpy)r)
# --- stand-in for the missing third-party Bio.pairwise2 (NOT reference code) ---
def _globalms(a, b, match, mismatch, open_, extend):
    n, m = len(a), len(b)
    S = [[0]*(m+1) for _ in range(n+1)]
    for i in range(1, n+1): S[i][0] = -i
    for j in range(1, m+1): S[0][j] = -j
    for i in range(1, n+1):
        for j in range(1, m+1):
            S[i][j] = max(S[i-1][j-1] + (match if a[i-1] == b[j-1] else mismatch), S[i-1][j] + extend, S[i][j-1] + extend)
    # depth-first traceback: gap in a, then diagonal, then gap in b; no gap-in-a right after a gap-in-b
    out = None
    stack = [(n, m, 0, False, "", "")]
    while stack:
        i, j, opt, colgap, ra, rb = stack.pop()
        if i == 0 and j == 0:
            out = (ra[::-1], rb[::-1]); break
        for o in (2, 1, 0):
            if o < opt: continue
            cur = S[i][j]
            if o == 0 and j > 0 and cur == S[i][j-1] + extend and not colgap:
                stack.append((i, j-1, 0, False, ra + "-", rb + b[j-1]))
            elif o == 1 and i > 0 and j > 0 and cur == S[i-1][j-1] + (match if a[i-1] == b[j-1] else mismatch):
                stack.append((i-1, j-1, 0, False, ra + a[i-1], rb + b[j-1]))
            elif o == 2 and i > 0 and cur == S[i-1][j] + extend:
                stack.append((i-1, j, 0, True, ra + a[i-1], rb + "-"))
    return [(out[0], out[1], S[n][m], 0, len(out[0]))]
Bio = types.ModuleType("Bio"); pw = types.ModuleType("Bio.pairwise2")
pw.align = types.SimpleNamespace(globalms=_globalms); pw.format_alignment = lambda *a, **k: ""
Bio.pairwise2 = pw; sys.modules["Bio"] = Bio; sys.modules["Bio.pairwise2"] = pw
spec = importlib.util.spec_from_file_location("jasper_ref", %(ref)r + "/jasper.py")
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
args = json.loads(sys.argv[1])
if args.get("debug"):
    orig = m.iteration
    def wrapped(*a, **k):
        m.debug = True
        return orig(*a, **k)
    m.iteration = wrapped
# track rows appended by the >k branch (list-valued fields / pairs)
m.main(None, args["query"], args["k"], True, True, args["fout"], args["ff"], args["db"], args["thre"], args["passes"])
'''


def run(cmd, **kw):
    return subprocess.run(cmd, check=True, **kw)


def rc(s):
    return s[::-1].translate(str.maketrans("ACGTacgt", "TGCAtgca"))


def rand_seq(rng, n):
    return "".join(np.array(list("ACGT"))[rng.integers(0, 4, n)])


def sample_reads(rng, haps, cov, rl, err):
    """uniform reads from each haplotype in `haps` (list of (seq, weight)); substitution errors at rate err"""
    reads = []
    for seq, w in haps:
        L = len(seq)
        if L < rl:
            continue
        n = int(L * cov * w / rl)
        for _ in range(n):
            s = int(rng.integers(0, L - rl + 1))
            r = list(seq[s:s + rl])
            if err > 0:
                for p in np.nonzero(rng.random(rl) < err)[0]:
                    r[p] = "ACGT"[(("ACGT".index(r[p]) if r[p] in "ACGT" else 0) + int(rng.integers(1, 4))) % 4]
            r = "".join(r)
            if rng.random() < 0.5:
                r = rc(r)
            reads.append(r)
    order = rng.permutation(len(reads))
    return [reads[i] for i in order]


def write_reads(path, reads, fmt, rng):
    with open(path, "w", newline="") as f:
        for i, r in enumerate(reads):
            if fmt == "fq":
                f.write("@r%d\n%s\n+\n%s\n" % (i, r, "I" * len(r)))
            elif fmt == "fa":
                f.write(">r%d some description\n%s\n" % (i, r))
            elif fmt == "fa_multiline":
                f.write(">r%d\n" % i)
                w = int(rng.integers(17, 61))
                for a in range(0, len(r), w):
                    f.write(r[a:a + w] + "\n")
            elif fmt == "fq_crlf":
                f.write("@r%d\r\n%s\r\n+\r\n%s\r\n" % (i, r, "I" * len(r)))
            elif fmt == "fq_multiline":
                h = len(r) // 2
                f.write("@r%d\n%s\n%s\n+r%d\n%s\n%s\n" % (i, r[:h], r[h:], i, "I" * h, "@" * (len(r) - h)))
            else:
                raise ValueError(fmt)


def mutate(rng, truth, plan):
    """apply a list of edits (pos, kind, arg) given in TRUTH coordinates, right to left"""
    s = list(truth)
    for pos, kind, arg in sorted(plan, key=lambda x: -x[0]):
        if kind == "sub":
            s[pos] = "ACGT"[("ACGT".index(s[pos]) + arg) % 4]
        elif kind == "ins":
            s[pos:pos] = list(arg)
        elif kind == "del":
            del s[pos:pos + arg]
        elif kind == "set":
            s[pos:pos + len(arg)] = list(arg)
        elif kind == "lower":
            s[pos:pos + arg] = [c.lower() for c in s[pos:pos + arg]]
    return "".join(s)


def case_specs():
    """name -> dict(k, passes, thre, rl, cov, err, fmt, build(rng) -> (haps, chunks))"""
    specs = {}

    def add(name, **kw):
        specs[name] = kw

    # 1. simple sub / ins / del, well separated, k=25
    def b_simple(rng, k):
        t = rand_seq(rng, 4000)
        plan = [(300, "sub", 1), (700, "ins", "G"), (1100, "del", 1), (1500, "sub", 2), (1900, "ins", "T"),
                (2300, "del", 1), (2700, "sub", 3), (3100, "sub", 1), (3500, "del", 1)]
        return [(t, 1.0)], [("ctg1:0", mutate(rng, t, plan))]
    add("simple_k25", k=25, passes=2, thre=5, rl=100, cov=40, err=0.002, fmt="fq", build=b_simple)
    add("simple_k37", k=37, passes=2, thre=5, rl=150, cov=40, err=0.002, fmt="fq", build=b_simple)
    add("simple_k17_p1", k=17, passes=1, thre=4, rl=80, cov=40, err=0.002, fmt="fa", build=b_simple)
    add("simple_k31_p3", k=31, passes=3, thre=5, rl=120, cov=40, err=0.002, fmt="fa_multiline", build=b_simple)
    # k beyond 43: keys of more than 86 bits (jasper.sh -k takes any k, src/jasper.sh:89-92)
    add("simple_k45", k=45, passes=2, thre=4, rl=150, cov=40, err=0.002, fmt="fq", build=b_simple)
    add("simple_k63", k=63, passes=2, thre=3, rl=150, cov=50, err=0.001, fmt="fq", build=b_simple)

    # 2. homopolymer indels
    def b_homo(rng, k):
        t = list(rand_seq(rng, 5000))
        sites = [400, 900, 1400, 1900, 2400, 2900, 3400, 3900, 4400]
        runs = [5, 7, 9, 6, 8, 10, 4, 12, 6]
        for s, r in zip(sites, runs):
            b = "ACGT"[int(rng.integers(0, 4))]
            t[s:s + r] = [b] * r
        t = "".join(t)
        plan = []
        for q, (s, r) in enumerate(zip(sites, runs)):
            if q % 3 == 0:
                plan.append((s + 1, "ins", t[s]))          # one extra copy
            elif q % 3 == 1:
                plan.append((s + 1, "del", 1))             # one copy missing
            elif q % 3 == 2:
                plan.append((s + 1, "ins", t[s] * 2))      # two extra copies
        plan.append((4700, "del", 2))
        return [(t, 1.0)], [("hp:0", mutate(rng, t, plan))]
    add("homopolymer_k25", k=25, passes=2, thre=5, rl=100, cov=40, err=0.002, fmt="fq", build=b_homo)
    add("homopolymer_k21", k=21, passes=2, thre=5, rl=100, cov=40, err=0.002, fmt="fq_crlf", build=b_homo)

    # 3. diploid sites: reads from two haplotypes, assembly carries haplotype mixtures
    def b_dip(rng, k):
        t = rand_seq(rng, 5000)
        hplan = [(p, "sub", 1 + int(rng.integers(0, 3))) for p in range(350, 4800, 450)]
        h2 = mutate(rng, t, hplan)
        # assembly: switches haplotype inside pairs of close het sites -> a chimeric k-mer set
        plan = [(1000, "sub", 1), (1003, "sub", 2), (2000, "sub", 1), (2010, "sub", 1), (3000, "sub", 3), (3002, "del", 1)]
        return [(t, 0.5), (h2, 0.5)], [("dip:0", mutate(rng, t, plan))]
    add("diploid_k25", k=25, passes=2, thre=4, rl=100, cov=60, err=0.002, fmt="fq", build=b_dip)

    # 4. clustered errors (two or more within k) -> base_extension
    def b_cluster(rng, k):
        t = rand_seq(rng, 6000)
        plan = []
        for q, p in enumerate(range(400, 5600, 520)):
            d = [3, 8, 15, 20, 24, 30, 40, 11, 5, 18][q % 10]
            plan.append((p, "sub", 1))
            kind = ["sub", "ins", "del"][q % 3]
            plan.append((p + d, kind, {"sub": 2, "ins": "C", "del": 1}[kind]))
            if q % 4 == 0:
                plan.append((p + 2 * d, "sub", 3))
        return [(t, 1.0)], [("cl:0", mutate(rng, t, plan))]
    add("cluster_k25", k=25, passes=2, thre=5, rl=100, cov=40, err=0.002, fmt="fq", build=b_cluster)
    add("cluster_k37", k=37, passes=2, thre=5, rl=150, cov=40, err=0.002, fmt="fq", build=b_cluster)

    # 5. chunk edges, N runs, lower case, odd characters, tiny contigs, several chunks
    def b_edges(rng, k):
        t = rand_seq(rng, 3000)
        c1 = mutate(rng, t[:1000], [(5, "sub", 1), (400, "sub", 1), (1000 - k - 3, "sub", 2), (990, "sub", 1)])
        c2 = mutate(rng, t[1000:2000], [(200, "set", "N" * 30), (500, "lower", 200), (600, "sub", 1), (800, "set", "n"),
                                        (850, "set", "R"), (900, "set", "NNNNN"), (950, "sub", 2)])
        c3 = mutate(rng, t[2000:3000], [(k - 1, "sub", 1), (300, "ins", "A"), (700, "del", 1)])
        tiny = [("t0:0", t[100:100 + k - 2]), ("t1:0", t[200:200 + k]), ("t2:0", t[300:300 + k + 1]),
                ("t3:0", t[400:400 + 2 * k]), ("t4:0", mutate(rng, t[500:500 + 3 * k + 5], [(k + 3, "sub", 1)])), ("t5:0", "")]
        return [(t, 1.0)], [("e:0", c1), ("e:1000", c2), ("e:2000", c3)] + tiny
    add("edges_k25", k=25, passes=2, thre=5, rl=100, cov=40, err=0.002, fmt="fq_multiline", build=b_edges)
    add("edges_k19", k=19, passes=2, thre=5, rl=100, cov=40, err=0.002, fmt="fa", build=b_edges)

    # 6. rolling threshold: high-copy repeat with a low-frequency variant, then unique sequence
    def b_roll(rng, k):
        unit = rand_seq(rng, 700)
        left, mid, right = rand_seq(rng, 1200), rand_seq(rng, 1500), rand_seq(rng, 1200)
        asm = left + unit + mid + unit + right
        haps = [(asm, 1.0)] + [(rand_seq(rng, 150) + unit + rand_seq(rng, 150), 1.0) for _ in range(24)]
        # a variant of the unit present in the assembly only at low frequency in the reads
        v = mutate(rng, unit, [(300, "sub", 1)])
        haps.append((rand_seq(rng, 150) + v + rand_seq(rng, 150), 0.25))
        asm2 = left + v + mid + mutate(rng, unit, [(500, "sub", 2)]) + right
        asm2 = mutate(rng, asm2, [(600, "sub", 1), (3000, "del", 1)])
        return haps, [("roll:0", asm2)]
    add("rolling_k25", k=25, passes=2, thre=5, rl=100, cov=40, err=0.002, fmt="fq", build=b_roll)
    add("rolling_k37", k=37, passes=2, thre=5, rl=150, cov=40, err=0.002, fmt="fq", build=b_roll)

    # 7. coverage gaps (long bad runs that cannot be repaired) + dense random errors
    def b_gaps(rng, k):
        t = rand_seq(rng, 6000)
        reads_from = [(t[:2000], 1.0), (t[2300:4000], 1.0), (t[4050:], 1.0)]
        plan = [(int(p), ["sub", "ins", "del"][int(rng.integers(0, 3))], None) for p in sorted(rng.choice(5800, 40, replace=False) + 100)]
        plan = [(p, kd, {"sub": 1 + int(rng.integers(0, 3)), "ins": "ACGT"[int(rng.integers(0, 4))], "del": 1}[kd]) for p, kd, _ in plan]
        return reads_from, [("gap:0", mutate(rng, t, plan))]
    add("gaps_k25", k=25, passes=2, thre=5, rl=100, cov=40, err=0.003, fmt="fq", build=b_gaps)
    add("gaps_k37_p4", k=37, passes=4, thre=6, rl=150, cov=50, err=0.003, fmt="fq", build=b_gaps)
    return specs


def build_case(name, spec, outdir, seed):
    rng = np.random.default_rng(seed)
    k = spec["k"]
    haps, chunks = spec["build"](rng, k)
    reads = sample_reads(rng, haps, spec["cov"], spec["rl"], spec["err"])
    # a few degenerate reads: shorter than k, with N, lower case
    reads += ["ACGT", haps[0][0][:k - 1], haps[0][0][10:10 + k].lower(), haps[0][0][50:90] + "N" + haps[0][0][91:140]]
    cdir = os.path.join(outdir, name)
    if os.path.isdir(cdir):
        shutil.rmtree(cdir)
    os.makedirs(cdir)
    work = tempfile.mkdtemp(prefix="golden_")
    ext = "fq" if spec["fmt"].startswith("fq") else "fa"
    rpath = os.path.join(work, "reads." + ext)
    write_reads(rpath, reads, spec["fmt"], rng)
    with open(os.path.join(work, "batch.fa"), "w") as f:
        for nm, s in chunks:
            f.write(">%s extra words\n%s\n" % (nm, s))
    env = dict(os.environ, PATH=os.path.dirname(JF_BIN) + ":" + os.environ["PATH"])
    db = os.path.join(work, "db.jf")
    run([JF_BIN, "count", "-C", "-s", "100000", "-m", str(k), "-o", db, "-t", "2", rpath], env=env)
    histo = subprocess.run([JF_BIN, "histo", "-t", "2", db], check=True, capture_output=True, env=env).stdout
    dump = subprocess.run([JF_BIN, "dump", "-c", db], check=True, capture_output=True, env=env).stdout
    dump = b"".join(sorted(dump.splitlines(keepends=True)))
    open(os.path.join(work, "histo.csv"), "wb").write(histo)
    thr = subprocess.run([sys.executable, os.path.join(REF, "jellyfish.py"), os.path.join(work, "histo.csv")], capture_output=True)
    # polisher
    args = dict(query="batch.fa", k=k, fout="batch.fa.fix.csv", ff="batch.fa.fixed.fa.tmp", db="db.jf",
                thre=spec["thre"], passes=spec["passes"], debug=True)
    drv = os.path.join(work, "drv.py")
    open(drv, "w").write(DRIVER % dict(jfpy=JF_PY, ref=REF))
    p = subprocess.run([sys.executable, drv, json.dumps(args)], cwd=work, capture_output=True, text=True)
    P = spec["passes"]
    # ---- store inputs
    with gzip.GzipFile(os.path.join(cdir, "reads.%s.gz" % ext), "wb", mtime=0) as f:
        f.write(open(rpath, "rb").read())
    shutil.copy(os.path.join(work, "batch.fa"), os.path.join(cdir, "batch.fa"))
    # ---- store reference outputs
    with gzip.GzipFile(os.path.join(cdir, "dump.txt.gz"), "wb", mtime=0) as f:
        f.write(dump)
    open(os.path.join(cdir, "histo.csv"), "wb").write(histo)
    if name in ("simple_k25", "simple_k37"):      # the reference's own DB file, for the .jf reader (jasper.sh -j)
        shutil.copy(db, os.path.join(cdir, "db.jf"))
    meta = dict(k=k, passes=P, thre=spec["thre"], reads_format=spec["fmt"], seed=seed,
                jellyfish_py_exit=thr.returncode, jellyfish_py_stdout=thr.stdout.decode(),
                jasper_py_exit=p.returncode)
    branches = sorted({" ".join(ln.split()[:2]) for ln in p.stdout.splitlines()
                       if ln and not ln[0].isdigit() and not ln.startswith("(")})
    meta["debug_messages_seen"] = branches
    if p.returncode == 0:
        unpinned = []
        for it in range(P):
            src = os.path.join(work, "_iter%d_batch.fa.fix.csv" % it)
            data = open(src, "rb").read()
            open(os.path.join(cdir, "iter%d.fix.csv" % it), "wb").write(data)
        shutil.copy(os.path.join(work, "_iter%d_batch.fa.fixed.fa.tmp" % (P - 1)), os.path.join(cdir, "fixed.fa"))
        meta["qv0"] = open(os.path.join(work, "0qValCalcHelper.csv")).read()
        meta["qvP"] = open(os.path.join(work, "%dqValCalcHelper.csv" % P)).read()
        # rows produced by the >k branch: located through the debug trace ("Looking for a path" precedes them)
        meta["n_base_extension_calls"] = sum(1 for ln in p.stdout.splitlines() if ln.startswith("Looking for a path"))
        meta["n_base_extension_success"] = sum(1 for ln in p.stdout.splitlines() if ln.startswith("Success path"))
    else:
        meta["jasper_py_stdout_tail"] = p.stdout[-400:]
    json.dump(meta, open(os.path.join(cdir, "meta.json"), "w"), indent=1, sort_keys=True)
    shutil.rmtree(work)
    return meta


# The seed of every case, by NAME (what its committed fixture was generated with; it is also in the case's meta.json).  A seed
# derived from a case's position in case_specs() changes for every case behind a newly inserted one: round 2 inserted two cases
# and the generator silently stopped reproducing eleven of the committed fixtures.  A new case gets a new entry here.
CASE_SEEDS = {
    "simple_k25": 1000, "simple_k37": 1001, "simple_k17_p1": 1002, "simple_k31_p3": 1003, "simple_k45": 1004, "simple_k63": 1005,
    "homopolymer_k25": 1004, "homopolymer_k21": 1005, "diploid_k25": 1006, "cluster_k25": 1007, "cluster_k37": 1008,
    "edges_k25": 1009, "edges_k19": 1010, "rolling_k25": 1011, "rolling_k37": 1012, "gaps_k25": 1013, "gaps_k37_p4": 1014,
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "cases"))
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    specs = case_specs()
    base = os.path.dirname(os.path.abspath(a.out))      # mer_kats.json and e2e/ live next to the cases directory
    if a.only in (None, "kats") and os.environ.get("GOLDEN_KATS", "1") == "1":
        make_mer_kats(os.path.join(base, "mer_kats.json"))
    if a.only in (None, "e2e") and os.environ.get("GOLDEN_E2E", "1") == "1":
        m = make_e2e(os.path.join(base, "e2e"))
        print("e2e exit", m["exit"], m["batches"], m["stdout"])
        m = make_e2e(os.path.join(base, "e2e_k45"), k=45, seed=78, rl=150)
        print("e2e_k45 exit", m["exit"], m["batches"], m["stdout"])
    for name, spec in specs.items():
        if a.only and a.only != name:
            continue
        if name not in CASE_SEEDS:
            sys.exit("case %r has no entry in CASE_SEEDS" % name)
        meta = build_case(name, spec, a.out, seed=CASE_SEEDS[name])
        print(name, "exit", meta["jasper_py_exit"], "qv0", meta.get("qv0", "").strip(), "qvP", meta.get("qvP", "").strip(),
              "ext", meta.get("n_base_extension_calls"), meta.get("n_base_extension_success"))
        print("   ", meta["debug_messages_seen"])




def make_mer_kats(out_path):
    """known answers of the reference's own encoder: str(MerDNA(s)) and str(MerDNA(s).get_canonical()) from the real
    SWIG binding (JF::swig/mer_dna.i) for strings with lower case, N at various offsets, short and empty input."""
    code = r'''
import sys, json
sys.path.insert(0, %r)
import dna_jellyfish as jf
import random
random.seed(5)
out = []
for k in (5, 17, 25, 31, 32, 33, 37, 48, 63):
    jf.MerDNA.k(k)
    cases = ["", "A", "GT", "N", "ACGTN", "acgtacgt"]
    for _ in range(12):
        n = random.choice([k, k, k, k - 1, k + 3, k // 2, 2 * k])
        s = "".join(random.choice("ACGT") for _ in range(n))
        r = random.random()
        if r < 0.25:
            s = s.lower()
        elif r < 0.5 and n > 2:
            p = random.randrange(n)
            s = s[:p] + random.choice("NnRX-") + s[p + 1:]
        elif r < 0.6:
            s = "".join(random.choice("ACGTacgt") for _ in range(n))
        cases.append(s)
    for s in cases:
        m = jf.MerDNA(s)
        c = m.get_canonical()
        out.append(dict(k=k, s=s, mer=str(m), canonical=str(c)))
json.dump(out, open(sys.argv[1], "w"), indent=0)
''' % JF_PY
    subprocess.run([sys.executable, "-c", code, out_path], check=True)




def make_e2e(outdir, k=25, seed=77, rl=100):
    """one run of the REAL src/jasper.sh (bash + perl + jellyfish + jasper.py) -> tests/golden/e2e/* (k = 25) and
    tests/golden/e2e_k45/* (k = 45: keys of 90 bits through the whole driver, incl. mer_counts45.jf).
    QV lines are not captured: `bc` is not installed in the build container."""
    import glob as _glob
    rng = np.random.default_rng(seed)
    contigs = [("ctgA some description", rand_seq(rng, 9000)), ("ctgB", rand_seq(rng, 5000)), ("ctgC:x", rand_seq(rng, 2500))]
    truth = {n.split()[0]: s for n, s in contigs}
    asm = {}
    for n, s in truth.items():
        L = len(s)
        plan = [(int(p), ["sub", "ins", "del"][int(rng.integers(0, 3))], None) for p in sorted(rng.choice(L - 200, max(3, L // 700), replace=False) + 100)]
        plan = [(p, kd, {"sub": 1 + int(rng.integers(0, 3)), "ins": "ACGT"[int(rng.integers(0, 4))], "del": 1}[kd]) for p, kd, _ in plan]
        asm[n] = mutate(rng, s, plan)
    reads = sample_reads(rng, [(s, 1.0) for s in truth.values()], 40, rl, 0.002)
    work = tempfile.mkdtemp(prefix="golden_e2e_")
    half = len(reads) // 2
    write_reads(os.path.join(work, "r1.fq"), reads[:half], "fq", rng)
    write_reads(os.path.join(work, "r2.fq"), reads[half:], "fq", rng)
    with open(os.path.join(work, "asm.fa"), "w") as f:
        for (hdr, _), (n, s) in zip(contigs, asm.items()):
            f.write(">%s\n" % hdr)
            for a in range(0, len(s), 70):
                f.write(s[a:a + 70] + "\n")
    # PYTHONPATH must be ONE directory holding jellyfish.py, dna_jellyfish and Bio (src/jasper.sh:115)
    pp = os.path.join(work, "pp")
    os.makedirs(os.path.join(pp, "Bio"))
    for fn in ("dna_jellyfish.py", "_dna_jellyfish.so"):
        shutil.copy(os.path.join(JF_PY, fn), pp)
    shutil.copy(os.path.join(REF, "jellyfish.py"), pp)
    open(os.path.join(pp, "Bio", "__init__.py"), "w").write("")
    stub = DRIVER.split("# --- stand-in")[1].split("Bio = types.ModuleType")[0]
    open(os.path.join(pp, "Bio", "pairwise2.py"), "w").write(
        "# stand-in" + stub + "\nimport types\nalign = types.SimpleNamespace(globalms=_globalms)\ndef format_alignment(*a, **k): return ''\n")
    # jasper.sh wants jasper.py executable next to itself: run a private copy of the two scripts from a temp bin dir
    bindir = os.path.join(work, "bin")
    os.makedirs(bindir)
    for fn in ("jasper.sh", "jasper.py", "jellyfish.py"):
        shutil.copy(os.path.join(REF, fn), bindir)
        os.chmod(os.path.join(bindir, fn), 0o755)
    env = dict(os.environ, PATH=bindir + ":" + os.path.dirname(JF_BIN) + ":" + os.environ["PATH"], PYTHONPATH=pp)
    run_dir = os.path.join(work, "run")
    os.makedirs(run_dir)
    for fn in ("r1.fq", "r2.fq", "asm.fa"):
        shutil.copy(os.path.join(work, fn), run_dir)
    p = subprocess.run(["bash", os.path.join(bindir, "jasper.sh"), "-r", "r1.fq r2.fq", "-a", "asm.fa", "-k", str(k), "-t", "4", "-p", "2", "-d"],
                       cwd=run_dir, env=env, capture_output=True, text=True)
    if os.path.isdir(outdir):
        shutil.rmtree(outdir)
    os.makedirs(outdir)
    for fn in ("asm.fa",):
        shutil.copy(os.path.join(run_dir, fn), outdir)
    for fn in ("r1.fq", "r2.fq"):
        with gzip.GzipFile(os.path.join(outdir, fn + ".gz"), "wb", mtime=0) as f:
            f.write(open(os.path.join(run_dir, fn), "rb").read())
    for fn in ("asm.fa.fixes.csv", "jfhisto%d.csv" % k, "threshold.txt"):
        shutil.copy(os.path.join(run_dir, fn), outdir)
    # the join step prints contigs in perl's hash order, which changes from run to run: the fixture keeps the records
    # sorted by name (the tests compare per record anyway), so that regenerating does not dirty it
    recs = open(os.path.join(run_dir, "asm.fa.polished.fasta")).read().split(">")[1:]
    with open(os.path.join(outdir, "asm.fa.polished.fasta"), "w") as f:
        f.write("".join(">" + r for r in sorted(recs)))
    batches = {}
    for bf in sorted(_glob.glob(os.path.join(run_dir, "asm.fa.batch.*.fa"))):
        batches[os.path.basename(bf)] = [ln.strip() for ln in open(bf) if ln.startswith(">")]
    log_lines = [re_sub_date(ln) for ln in p.stdout.splitlines()]
    meta = dict(k=k, threads=4, passes=2, exit=p.returncode, batches=batches, stdout=log_lines,
                sentinels=sorted(os.path.basename(x) for x in _glob.glob(os.path.join(run_dir, "jasper.*.success"))))
    json.dump(meta, open(os.path.join(outdir, "meta.json"), "w"), indent=1, sort_keys=True)
    shutil.rmtree(work)
    return meta


def re_sub_date(line):
    import re
    return re.sub(r"^\[[^\]]*\]", "[DATE]", line)


if __name__ == "__main__":
    main()
