#!/usr/bin/env python3
"""long GPU fuzz run (HIP path vs oracle) with progress lines: python tools/fuzz_gpu.py SEED0 N [logfile]"""
import os, sys, tempfile, pathlib, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import test_gpu_fuzz as T
from jasper_amd import KmerTable, polisher
from oracle import oracle as O
import make_golden as G
import fuzz_vs_reference as F

seed0, n = int(sys.argv[1]), int(sys.argv[2])
log = open(sys.argv[3], "a") if len(sys.argv) > 3 else sys.stdout
tmp = pathlib.Path(tempfile.mkdtemp(prefix="fuzzgpu_"))
t0 = time.time()
bad = 0
for i, seed in enumerate(range(seed0, seed0 + n)):
    try:
        T._one(seed, KmerTable, polisher, O, G, F, tmp)
    except AssertionError as e:
        bad += 1
        import traceback
        tb = traceback.extract_tb(e.__traceback__)[-1]
        log.write("seed %d FAILED at test_gpu_fuzz.py:%d (%s): %s\n" % (seed, tb.lineno, tb.line, str(e)[:300]))
        for again in range(3):          # the same case again in this very process: does it fail every time?
            try:
                T._one(seed, KmerTable, polisher, O, G, F, tmp)
                log.write("   again %d: passes\n" % again)
            except AssertionError as e2:
                tb2 = traceback.extract_tb(e2.__traceback__)[-1]
                log.write("   again %d: FAILS at line %d (%s)\n" % (again, tb2.lineno, tb2.line))
        log.flush()
    for f in tmp.iterdir():
        f.unlink()
    if (i + 1) % 250 == 0:
        log.write("%d cases, %d failing, %.0f s\n" % (i + 1, bad, time.time() - t0))
        log.flush()
log.write("done: seeds %d..%d, %d failing\n" % (seed0, seed0 + n - 1, bad))
log.flush()
sys.exit(1 if bad else 0)
