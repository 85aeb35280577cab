#!/usr/bin/env python3
"""GPU box: a few seeds of the differential fuzz over and over in one process (a failure that comes and goes): python tools/fuzz_repeat.py SEED0 NSEEDS REPEATS"""
import os, sys, tempfile, pathlib, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import test_gpu_fuzz as T
from jasper_amd import KmerTable, polisher
from oracle import oracle as O
import make_golden as G
import fuzz_vs_reference as F
seed0, ns, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
# which side moves when a run fails?  Both polishers' last outputs are kept and compared with what the SAME seed gave the first time
last = {}
_gpu, _ora = polisher.polish_batch, O.OracleDB.polish_batch
def gpu_polish(*a, **k):
    out = _gpu(*a, **k); last["gpu"] = (out[0], out[2]); return out
def ora_polish(self, *a, **k):
    out = _ora(self, *a, **k); last["oracle"] = (out[0], out[2]); return out
polisher.polish_batch = gpu_polish
O.OracleDB.polish_batch = ora_polish
first = {}
tmp = pathlib.Path(tempfile.mkdtemp(prefix="fuzzrep_"))
bad, t0 = 0, time.time()
for r in range(reps):
    for seed in range(seed0, seed0 + ns):
        try:
            T._one(seed, KmerTable, polisher, O, G, F, tmp)
        except AssertionError as e:
            bad += 1
            tb = traceback.extract_tb(e.__traceback__)[-1]
            print("rep %d seed %d FAILED at line %d (%s)" % (r, seed, tb.lineno, tb.line), flush=True)
            if seed in first:
                for side in ("gpu", "oracle"):
                    same = last.get(side) == first[seed].get(side)
                    print("   %s: %s its first run's result; qv %s" % (side, "EQUALS" if same else "DIFFERS FROM", last.get(side, (None, None))[1]), flush=True)
                    if not same and side in last and side in first[seed]:
                        for ci, (a, b) in enumerate(zip(last[side][0], first[seed][side][0])):
                            if a != b:
                                j = next((q for q in range(min(len(a), len(b))) if a[q] != b[q]), min(len(a), len(b)))
                                print("      chunk %d: lens %d / %d, first difference at %d: %s | %s" % (ci, len(a), len(b), j, a[max(0, j - 25):j + 35], b[max(0, j - 25):j + 35]), flush=True)
        if seed not in first and "gpu" in last and "oracle" in last:
            first[seed] = dict(last)
        last.clear()
        for f in tmp.iterdir():
            f.unlink()
    if (r + 1) % 100 == 0:
        print("%d repeats, %d failing, %.0f s" % (r + 1, bad, time.time() - t0), flush=True)
print("done: %d runs, %d failing" % (reps * ns, bad))
