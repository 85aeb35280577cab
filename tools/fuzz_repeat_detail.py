#!/usr/bin/env python3
"""GPU box: a few fuzz cases in turn, over and over, a fresh table per run (as tests/test_gpu_fuzz.py makes them); every run whose
polishing differs from the oracle is described (which chunk, where, qv, records): python tools/fuzz_repeat_detail.py SEED0 NSEEDS REPEATS"""
import os, sys, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from jasper_amd import KmerTable, polisher
from oracle import oracle as O
import make_golden as G
import fuzz_vs_reference as F
seed0, ns, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
tmp = pathlib.Path(tempfile.mkdtemp(prefix="fuzzrd_"))
cases = []
for seed in range(seed0, seed0 + ns):
    rng, spec, haps, chunks = F.random_case(seed)
    k = spec["k"]
    reads = G.sample_reads(rng, haps, spec["cov"], spec["rl"], spec["err"]) or [haps[0][0][:spec["rl"]]]
    ext = "fq" if spec["fmt"].startswith("fq") else "fa"
    rpath = str(tmp / ("reads%d.%s" % (seed, ext)))
    G.write_reads(rpath, reads, spec["fmt"], rng)
    odb = O.OracleDB(k); odb.count_text(open(rpath, "rb").read())
    names = [c[0] for c in chunks]; seqs = [c[1] for c in chunks]
    try:
        fixed_o, rows_o, qv_o, _ = odb.polish_batch(names, seqs, spec["thre"], spec["passes"])
    except RuntimeError:
        fixed_o = None
    print(seed, spec, "chunks", [len(s) for s in seqs], "oracle", "raises" if fixed_o is None else qv_o, flush=True)
    if fixed_o is not None:
        cases.append((seed, spec, rpath, seqs, fixed_o, qv_o))
bad = 0
for r in range(reps):
    for seed, spec, rpath, seqs, fixed_o, qv_o in cases:
        if os.environ.get("LIKE_THE_TEST"):            # everything tests/test_gpu_fuzz.py:_one does before it polishes
            rng, spec2, haps, chunks = F.random_case(seed)
            reads = G.sample_reads(rng, haps, spec2["cov"], spec2["rl"], spec2["err"]) or [haps[0][0][:spec2["rl"]]]
            os.remove(rpath)
            G.write_reads(rpath, reads, spec2["fmt"], rng)
            odb = O.OracleDB(spec["k"]); odb.count_text(open(rpath, "rb").read())
            items = list(odb.items())
        t = KmerTable(spec["k"], min_slots=1 << 16); t.count_files([rpath])
        if os.environ.get("LIKE_THE_TEST"):
            assert t.info()["distinct"] == len(items) and t.histogram() == odb.histo()
            sample = items[:: max(1, len(items) // 200)]
            assert t.lookup([km for km, _ in sample]) == [min(c, 0xFFFFFFFF) for _, c in sample]
            odb.polish_batch([c[0] for c in chunks], seqs, spec["thre"], spec["passes"])
        res = t.polish_batch(seqs, spec["thre"], spec["passes"])
        fixed = res.seqs
        if fixed != fixed_o or res.qv != qv_o:
            bad += 1
            print("run %d seed %d DIFFERS: qv %s (oracle %s), %d records, segments %d respeculated %d retried %s" % (r, seed, res.qv, qv_o, res.n_records, res.segments, res.respeculated, res.retried))
            for i, (a, b) in enumerate(zip(fixed, fixed_o)):
                if a != b:
                    j = next((q for q in range(min(len(a), len(b))) if a[q] != b[q]), min(len(a), len(b)))
                    print("   chunk %d: lens %d / %d, first difference at %d: ...%s | ...%s" % (i, len(a), len(b), j, a[max(0, j - 20):j + 40], b[max(0, j - 20):j + 40]))
            print("   records:", [(x["chunk"], x["pass_"], x["seqno"], x["kind"], x["index"], x["newc"], x["oldc"], x["rep"]) for x in res.records][:50], flush=True)
        if r == 0:
            print("seed %d run 0: segments %d, chunks redone unsegmented %d, retried %s, records %d" % (seed, res.segments, res.respeculated, res.retried, res.n_records), flush=True)
        del res
        t.close()
print("done: %d runs, %d differing" % (reps * len(cases), bad))
