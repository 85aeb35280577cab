#!/usr/bin/env python3
"""the bc emulation of jasper_amd/qv.py (series, as libmath.b) against correctly rounded ln / exp truncated where bc truncates,
over N random (bad, total, k) triples: how many Q strings differ, and by how much.   python tools/qv_compare.py [N=100000]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jasper_amd import qv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
rng = random.Random(12345)
t0 = time.time()
differ, worst, by = 0, 0.0, {}
for i in range(N):
    total = rng.randint(1000, 4 * 10 ** 9)
    bad = int(total * 10 ** rng.uniform(-7.5, -0.3))
    k = rng.choice([17, 21, 25, 31, 37, 45, 63])
    a, b = qv.q_value(bad, total, k), qv.q_value_exact(bad, total, k)
    if a != b:
        differ += 1
        d = round(abs(float(a) - float(b)) * 1e5) if "Inf" not in (a, b) else -1
        by[d] = by.get(d, 0) + 1
        worst = max(worst, d)
print("%d triples in %.0f s: %d Q strings differ (%.4f %%); differences in units of the fifth decimal: %s" % (N, time.time() - t0, differ, 100.0 * differ / N, dict(sorted(by.items()))))
