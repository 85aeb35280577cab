#!/usr/bin/env python3
"""GPU box: ONE case of the differential fuzz (tests/test_gpu_fuzz.py) with every check reported by itself: python tools/fuzz_one.py SEED"""
import os, sys, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from jasper_amd import KmerTable, polisher
from oracle import oracle as O
import make_golden as G
import fuzz_vs_reference as F
seed = int(sys.argv[1])
tmp = pathlib.Path(tempfile.mkdtemp(prefix="fuzz1_"))
rng, spec, haps, chunks = F.random_case(seed)
k = spec["k"]
reads = G.sample_reads(rng, haps, spec["cov"], spec["rl"], spec["err"]) or [haps[0][0][:spec["rl"]]]
ext = "fq" if spec["fmt"].startswith("fq") else "fa"
rpath = str(tmp / ("reads%d.%s" % (seed, ext)))
G.write_reads(rpath, reads, spec["fmt"], rng)
text = open(rpath, "rb").read()
print(spec, len(reads), "reads", len(text), "bytes of text; chunks", [len(c[1]) for c in chunks])
odb = O.OracleDB(k); odb.count_text(text)
t = KmerTable(k, min_slots=1 << 16); t.count_files([rpath])
items = list(odb.items())
print("distinct", t.info()["distinct"], len(items), "histogram equal", t.histogram() == odb.histo(), "ingest", t.last_ingest() if hasattr(t, "last_ingest") else None)
sample = items[:: max(1, len(items) // 200)]
print("lookups equal", t.lookup([km for km, _ in sample]) == [min(c, 0xFFFFFFFF) for _, c in sample])
names = [c[0] for c in chunks]; seqs = [c[1] for c in chunks]
try:
    fixed_o, rows_o, qv_o, _ = odb.polish_batch(names, seqs, spec["thre"], spec["passes"]); ok_o = True
except RuntimeError as e:
    ok_o = False; print("oracle raised", e)
try:
    fixed, rows, qv, _ = polisher.polish_batch(t, names, seqs, spec["thre"], spec["passes"]); ok = True
except BaseException as e:
    ok = False; print("gpu raised", repr(e)[:300])
print("ok", ok, "ok_o", ok_o)
if ok and ok_o:
    print("qv", qv, qv_o, "fixed equal", fixed == fixed_o)
    for i, (a, b) in enumerate(zip(fixed, fixed_o)):
        if a != b:
            j = next((q for q in range(min(len(a), len(b))) if a[q] != b[q]), min(len(a), len(b)))
            print(" chunk", i, "lens", len(a), len(b), "first difference at", j, a[max(0, j - 30):j + 30], "|", b[max(0, j - 30):j + 30])
    for it in range(spec["passes"]):
        print(" pass", it, "csv equal", polisher.fix_csv_text(rows[it]) == "Contig Base_coord Original Mutation\r\n" + rows_o[it])
