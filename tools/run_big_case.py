#!/usr/bin/env python3
"""GPU box: one of the large fixtures of tests/golden/fullsize_*.json through the drop-in CLI, with a progress line every minute
(what `JASPER_TEST_BIG=1 pytest tests/test_gpu_cli_fullsize.py` checks, for runs whose input alone takes minutes to write).
   python tools/run_big_case.py [fixture name, default fullsize_cfg4_share] [scratch dir, default /tmp]"""
import json, os, re, shutil, subprocess, sys, tempfile, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jasper_amd import synth, qv
name = sys.argv[1] if len(sys.argv) > 1 else "fullsize_cfg4_share"
base = sys.argv[2] if len(sys.argv) > 2 else "/tmp"
ref = json.load(open(os.path.join(ROOT, "tests", "golden", name + ".json")))
d = tempfile.mkdtemp(prefix="jasper_big_", dir=base)
stop = False
def beat():
    t0 = time.time()
    while not stop:
        time.sleep(30)
        print("[%4.0f s] working (%s)" % (time.time() - t0, ", ".join("%s %.1f GB" % (f, os.path.getsize(os.path.join(d, f)) / 1e9) for f in sorted(os.listdir(d)) if os.path.getsize(os.path.join(d, f)) > 1e8)[:200]), flush=True)
threading.Thread(target=beat, daemon=True).start()
try:
    t0 = time.time()
    nreads, asm_len = synth.write_cli_inputs(d, ref["genome_mb"], ref["seed"], coverage=ref.get("coverage", 30), contigs=ref.get("contigs", 1), populations=ref.get("populations", 1))
    print("inputs: %d reads, %d assembly bases, %.0f s" % (nreads, asm_len, time.time() - t0), flush=True)
    assert nreads == ref["reads"] and asm_len == ref["assembly_bases"]
    files = synth.read_files(ref.get("populations", 1))
    args = [sys.executable, "-m", "jasper_amd.cli", "-r", " ".join(files), "-a", "asm.fa", "-k", str(ref["k"]), "-t", str(ref["threads"]), "-p", str(ref["passes"])]
    inputs = set(os.listdir(d))
    for rep in range(int(os.environ.get("JASPER_BIG_REPEAT", "1"))):       # (a box's first run pays for device memory that was never handed out before)
        for fn in set(os.listdir(d)) - inputs:
            os.remove(os.path.join(d, fn))
        t1 = time.time()
        p = subprocess.run(args, cwd=d, env=dict(os.environ, PYTHONPATH=ROOT, JASPER_AMD_TIMING="1", JASPER_AMD_NO_JF="1", JASPER_COUNT_DEBUG="1"), capture_output=True, text=True)
        wall = time.time() - t1
        print("run %d: exit %d, wall %.2f s; %s" % (rep, p.returncode, wall, ", ".join("%s %s" % (a.strip(), b) for a, b in re.findall(r"\[timing\] (.*?)\s+([0-9.]+) s", p.stderr))), flush=True)
    print(p.stdout[-1500:])
    print("\n".join(l for l in p.stderr.splitlines() if not l.startswith("[polish]"))[-6000:])
    print("exit %d, wall %.1f s (reference: %.1f s on %s)" % (p.returncode, wall, ref["reference_wall_seconds"], ref["host"]), flush=True)
    got = synth.output_digests(d, k=ref["k"])
    keys = ("threshold", "jfhisto_sha256", "polished_bases", "polished_fasta_sha256", "fixes_csv_lines", "fixes_csv_sha256")
    for k in keys:
        print("  %-24s %s" % (k, "equal" if got[k] == ref[k] else "DIFFERENT: %r vs reference %r" % (got[k], ref[k])))
    m = re.search(r"^\[qv\] before (\d+) (\d+) after (\d+) (\d+)$", p.stderr, re.M)
    sums = [int(x) for x in m.groups()] if m else None
    print("  qv sums %s: %s (reference %s %s)" % ("equal" if sums and sums[:2] == ref["qv_before"] and sums[2:] == ref["qv_after"] else "DIFFERENT", sums, ref["qv_before"], ref["qv_after"]))
    ok = p.returncode == 0 and all(got[k] == ref[k] for k in keys) and sums and sums[:2] == ref["qv_before"] and sums[2:] == ref["qv_after"]
    print(json.dumps({"fixture": name, "drop_in_wall_seconds": round(wall, 2), "reference_wall_seconds": ref["reference_wall_seconds"], "all_equal": bool(ok),
                      "stage_seconds": {a: float(b) for a, b in re.findall(r"\[timing\] (.*?)\s+([0-9.]+) s", p.stderr)}}))
    sys.exit(0 if ok else 1)
finally:
    stop = True
    shutil.rmtree(d, ignore_errors=True)
