#!/usr/bin/env python3
"""full-size check of the database written by a multi-GPU driver run: 2 ranks (one GPU, gloo) run configs[1] and leave
mer_counts37.jf behind; a one-GPU run then REUSES that file (src/jasper.sh:171-173) and must arrive at the real reference's
digests (tests/golden/fullsize_cfg2.json).  python tools/check_multi_jf.py [workdir]"""
import json, os, re, socket, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jasper_amd import synth
ref = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize_cfg2.json")))
d = sys.argv[1] if len(sys.argv) > 1 else tempfile.mkdtemp(prefix="mjf_")
os.makedirs(d, exist_ok=True)
synth.write_cli_inputs(d, ref["genome_mb"], ref["seed"])
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
args = ["-r", "reads.fq", "-a", "asm.fa", "-k", str(ref["k"]), "-t", str(ref["threads"]), "-p", str(ref["passes"])]
env = dict(os.environ, PYTHONPATH=ROOT, JASPER_AMD_DIST_BACKEND="gloo", JASPER_AMD_ONE_GPU="1", JASPER_AMD_TIMING="1")
t0 = time.perf_counter()
p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                    "-m", "jasper_amd.cli"] + args, cwd=d, env=env, capture_output=True, text=True)
print("2 ranks: rc %d, %.1f s" % (p.returncode, time.perf_counter() - t0)); print("\n".join(l for l in p.stderr.splitlines() if "[timing]" in l))
assert p.returncode == 0, p.stdout + p.stderr
got = synth.output_digests(d, k=ref["k"])
keys = ("threshold", "jfhisto_sha256", "polished_bases", "polished_fasta_sha256", "fixes_csv_lines", "fixes_csv_sha256")
assert all(got[k] == ref[k] for k in keys), got
print("mer_counts37.jf: %d bytes" % os.path.getsize(os.path.join(d, "mer_counts37.jf")))
for fn in os.listdir(d):
    if re.match(r"jasper\..*\.success$", fn) or fn.endswith(".polished.fasta") or fn.endswith(".fixes.csv") or fn.startswith("jfhisto") or fn == "threshold.txt":
        os.remove(os.path.join(d, fn))
t0 = time.perf_counter()
p = subprocess.run([sys.executable, "-m", "jasper_amd.cli"] + args, cwd=d, env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True)
print("1 GPU reusing the file: rc %d, %.1f s" % (p.returncode, time.perf_counter() - t0))
assert p.returncode == 0 and "Using existing jellyfish database mer_counts37.jf" in p.stdout, p.stdout + p.stderr
got = synth.output_digests(d, k=ref["k"])
assert all(got[k] == ref[k] for k in keys), got
print("OK: digests of both runs == the real reference's")
