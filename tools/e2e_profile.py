#!/usr/bin/env python3
"""GPU box: where the wall time of the drop-in goes (configs[1] files -> files): interpreter start, the CLI's own stage marks, exit.
   python tools/e2e_profile.py [runs] [ENV=VALUE ...]"""
import os, re, subprocess, sys, tempfile, time, json, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jasper_amd import synth
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
extra = dict(a.split("=", 1) for a in sys.argv[2:])
ref = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize_cfg2_t16.json")))
d = tempfile.mkdtemp(prefix="jasper_e2e_", dir="/tmp")
try:
    synth.write_cli_inputs(d, ref["genome_mb"], ref["seed"], coverage=ref["coverage"])
    args = [sys.executable, "-m", "jasper_amd.cli", "-r", "reads.fq", "-a", "asm.fa", "-k", "37", "-t", "16", "-p", "2"]
    for r in range(runs):
        for fn in os.listdir(d):
            if fn not in ("reads.fq", "asm.fa"):
                os.remove(os.path.join(d, fn))
        time.sleep(3.0)
        t0 = time.time()
        p = subprocess.run(args, cwd=d, env=dict(os.environ, PYTHONPATH=ROOT, JASPER_AMD_TIMING="1", JASPER_AMD_NO_JF="1", **extra), capture_output=True, text=True)
        t1 = time.time()
        absm = {m.group(1): float(m.group(2)) for m in re.finditer(r"\[timing-abs\] run\(\) (\w+) at ([0-9.]+)", p.stderr)}
        marks = re.findall(r"\[timing\] (.*?)\s+([0-9.]+) s", p.stderr)
        got = synth.output_digests(d, k=37)
        ok = all(got[k] == ref[k] for k in ("threshold", "jfhisto_sha256", "polished_fasta_sha256", "fixes_csv_sha256"))
        print("run %d: wall %.3f s = start-up %.3f + run() %.3f + exit %.3f | %s | outputs equal reference: %s" % (
            r, t1 - t0, absm.get("entered", t0) - t0, absm.get("returned", t1) - absm.get("entered", t0), t1 - absm.get("returned", t1),
            ", ".join("%s %s" % (a, b) for a, b in marks), ok), flush=True)
        if p.returncode or "JASPER_COUNT_DEBUG" in extra:
            print(p.stderr[-3000:])
finally:
    shutil.rmtree(d, ignore_errors=True)
