#!/usr/bin/env python3
"""tests/golden/qv_kats.json: internal known answers for the QV lines (src/jasper.sh:239-256).

  * `bc_constants`: what `bc -l` prints for a few expressions (values every GNU bc prints; they pin the emulation's l / e / sqrt)
  * `fixtures`: (bad, total) before and after polishing of the e2e fixtures -- computed by the CPU ORACLE on the fixture's files,
    split like src/jasper.sh:132-156 -- and the Q strings jasper_amd.qv prints for them; the GPU CLI tests compare their whole
    'Q value = ...' lines with these instead of stripping the digits
  * `triples`: (bad, total, k) -> Q for a spread of magnitudes (regression pins of the emulation itself)
The reference's own log says 'Inf' for every Q (bc is not installed where it ran), so none of this comes from the reference:
the digits are PARITY UNPINNED (jasper_amd/qv.py)."""
import gzip, json, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jasper_amd import cli, qv, synth
from oracle import oracle as O

out = {"bc_constants": {"scale=20; l(2)": ".69314718055994530941", "scale=20; e(1)": "2.71828182845904523536", "scale=20; sqrt(2)": "1.41421356237309504880",
                        "scale=20; l(10)": "2.30258509299404568401"},
       "fixtures": {}, "triples": []}
G = os.path.join(ROOT, "tests", "golden")
for fx in ("e2e", "e2e_k45"):
    d = os.path.join(G, fx)
    meta = json.load(open(os.path.join(d, "meta.json")))
    k, P, T = meta["k"], meta["passes"], meta["threads"]
    db = O.OracleDB(k)
    for fn in ("r1.fq.gz", "r2.fq.gz"):
        db.count_text(gzip.open(os.path.join(d, fn)).read())
    thr = O.threshold(db.histo_rows() if hasattr(db, "histo_rows") else [(m, n) for m, n in enumerate(db.histo()) if n and m])
    contigs = cli.read_assembly(os.path.join(d, "asm.fa"))
    total_bases = sum(len(s) for _, s in contigs)
    bs = synth.jasper_batch_size(total_bases, T)
    qsum = [0, 0, 0, 0]
    for name, seq in contigs:
        for rec, a, b in synth.chunk_records(name, len(seq), bs):
            s = seq[a:b]
            s = s.decode() if isinstance(s, (bytes, bytearray)) else s
            _, _, q, _ = db.polish_batch([rec], [s], thr, P)
            for i in range(4):
                qsum[i] += q[i]
    out["fixtures"][fx] = {"k": k, "threshold": thr, "before": [qsum[0], qsum[1], qv.q_value(qsum[0], qsum[1], k)],
                           "after": [qsum[2], qsum[3], qv.q_value(qsum[2], qsum[3], k)]}
rng = random.Random(20261004)
for _ in range(200):
    total = rng.randint(1000, 4 * 10 ** 9)
    bad = int(total * 10 ** rng.uniform(-7.5, -0.3))
    k = rng.choice([17, 21, 25, 31, 37, 45, 63])
    out["triples"].append([bad, total, k, qv.q_value(bad, total, k)])
for t in ([0, 1000, 37], [5, 0, 37], [10, 10, 25], [1, 4 * 10 ** 9, 37], [999, 1000, 25]):
    out["triples"].append(t + [qv.q_value(*t)])
json.dump(out, open(os.path.join(G, "qv_kats.json"), "w"), indent=1)
print(json.dumps(out["fixtures"], indent=1))
