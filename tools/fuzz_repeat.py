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
        for f in tmp.iterdir():
            f.unlink()
    if (r + 1) % 100 == 0:
        print("%d repeats, %d failing, %.0f s" % (r + 1, bad, time.time() - t0), flush=True)
print("done: %d runs, %d failing" % (reps * ns, bad))
