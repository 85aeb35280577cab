#!/usr/bin/env python3
"""GPU box: the drop-in on a fragmented assembly (many short contigs): python tools/fragmented_cli_probe.py [n_contigs] [mean_len]"""
import os, shutil, subprocess, sys, tempfile, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from jasper_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
mean = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rng = np.random.default_rng(3)
G = n * mean
genome = synth.make_genome(rng, G)
reads = synth.make_reads_stream(rng, genome, 30, 150, 0.003).reshape(-1, 151)[:, :150]
d = tempfile.mkdtemp(prefix="jasper_frag_")
rec = np.empty((reads.shape[0], 307), dtype=np.uint8)
rec[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8); rec[:, 3:153] = reads; rec[:, 153:156] = np.frombuffer(b"\n+\n", dtype=np.uint8); rec[:, 156:306] = ord("I"); rec[:, 306] = ord("\n")
rec.tofile(os.path.join(d, "reads.fq"))
asm = synth.make_assembly(rng, genome).tobytes()
cuts = np.sort(rng.choice(np.arange(1, len(asm)), n - 1, replace=False))
with open(os.path.join(d, "asm.fa"), "wb") as f:
    a = 0
    for i, b in enumerate(list(cuts) + [len(asm)]):
        f.write(b">ctg%d some description\n" % i)
        s = asm[a:b]
        f.write(b"\n".join(s[j:j + 60] for j in range(0, len(s), 60)) + b"\n")
        a = b
env = dict(os.environ, PYTHONPATH=ROOT, JASPER_AMD_TIMING="1", JASPER_AMD_NO_JF="1")
t0 = time.perf_counter()
p = subprocess.run([sys.executable, "-m", "jasper_amd.cli", "-r", "reads.fq", "-a", "asm.fa", "-k", "37", "-t", "16", "-p", "2"], cwd=d, env=env, capture_output=True, text=True)
print("%d contigs, %.1f Mb: wall %.2f s rc %d" % (n, len(asm) / 1e6, time.perf_counter() - t0, p.returncode))
print("".join(ln + "\n" for ln in p.stderr.splitlines() if ln.startswith("[timing]")))
print(p.stdout[-600:])
if p.returncode:
    print(p.stderr[-1500:])
out = os.path.join(d, "asm.fa.polished.fasta")
if os.path.exists(out):
    txt = open(out).read()
    print("polished records:", txt.count(">"), "bases:", len(txt) - txt.count("\n") - sum(len(l) for l in txt.split("\n") if l.startswith(">")))
shutil.rmtree(d, ignore_errors=True)
