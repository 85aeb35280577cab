#!/usr/bin/env python3
"""end-to-end drop-in run at bench scale (GPU box): synthetic FASTQ + FASTA files on disk -> python -m jasper_amd.cli.
   python tools/bench_cli.py [genome_mb]   (files go to /tmp/jasper_cli_bench, removed afterwards)"""
import os, shutil, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from jasper_amd import synth

gmb = float(sys.argv[1]) if len(sys.argv) > 1 else 47.0
d = "/tmp/jasper_cli_bench"
shutil.rmtree(d, ignore_errors=True)
os.makedirs(d)
rng = np.random.default_rng(2)
t0 = time.perf_counter()
genome = synth.make_genome(rng, int(gmb * 1e6))
reads = synth.make_reads_stream(rng, genome, 30, 150, 0.003).reshape(-1, 151)[:, :150]
n = reads.shape[0]
rec = np.empty((n, 307), dtype=np.uint8)
rec[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8)
rec[:, 3:153] = reads
rec[:, 153:156] = np.frombuffer(b"\n+\n", dtype=np.uint8)
rec[:, 156:306] = ord("I")
rec[:, 306] = ord("\n")
rec.tofile(os.path.join(d, "reads.fq"))
asm = synth.make_assembly(rng, genome)
with open(os.path.join(d, "asm.fa"), "wb") as f:
    f.write(b">chr1\n")
    a = asm.tobytes()
    f.write(b"\n".join(a[i:i + 60] for i in range(0, len(a), 60)))
    f.write(b"\n")
print("inputs: %.2f GB FASTQ (%d reads), %.1f MB FASTA, generated in %.1f s" % (os.path.getsize(os.path.join(d, "reads.fq")) / 1e9, n, len(asm) / 1e6, time.perf_counter() - t0), flush=True)
del rec, reads, genome
env = dict(os.environ, PYTHONPATH=ROOT, JASPER_AMD_TIMING="1")
for label, extra in (("first run (writes mer_counts37.jf)", {}), ("second run in a clean directory, no .jf written", {"JASPER_AMD_NO_JF": "1"})):
    for fn in os.listdir(d):
        if fn not in ("reads.fq", "asm.fa"):
            os.remove(os.path.join(d, fn))
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, "-m", "jasper_amd.cli", "-r", "reads.fq", "-a", "asm.fa", "-k", "37", "-t", "16", "-p", "2"], cwd=d,
                       env=dict(env, **extra), capture_output=True, text=True)
    dt = time.perf_counter() - t0
    print("%s: exit %d, %.2f s wall -> %.1f Mbp/s end to end" % (label, p.returncode, dt, len(asm) / 1e6 / dt), flush=True)
    for ln in p.stdout.splitlines():
        print("   ", ln)
    print("".join("    " + ln + "\n" for ln in p.stderr.splitlines() if ln.startswith("[timing]")), end="")
    if p.returncode:
        print(p.stderr[-2000:])
shutil.rmtree(d, ignore_errors=True)
