import os, sys, shutil, subprocess, time, pstats
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import numpy as np
from jasper_amd import synth
d = "/tmp/jasper_cli_prof"
shutil.rmtree(d, ignore_errors=True); os.makedirs(d)
rng = np.random.default_rng(2)
genome = synth.make_genome(rng, 47_000_000)
reads = synth.make_reads_stream(rng, genome, 30, 150, 0.003).reshape(-1, 151)[:, :150]
n = reads.shape[0]
rec = np.empty((n, 307), dtype=np.uint8)
rec[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8); rec[:, 3:153] = reads; rec[:, 153:156] = np.frombuffer(b"\n+\n", dtype=np.uint8); rec[:, 156:306] = ord("I"); rec[:, 306] = ord("\n")
rec.tofile(os.path.join(d, "reads.fq"))
asm = synth.make_assembly(rng, genome)
with open(os.path.join(d, "asm.fa"), "wb") as f:
    f.write(b">chr1\n"); a = asm.tobytes(); f.write(b"\n".join(a[i:i + 60] for i in range(0, len(a), 60))); f.write(b"\n")
env = dict(os.environ, PYTHONPATH=ROOT, JASPER_AMD_TIMING="1", JASPER_AMD_NO_JF="1")
t0 = time.perf_counter()
p = subprocess.run([sys.executable, "-m", "cProfile", "-o", os.path.join(d, "prof.out"), "-m", "jasper_amd.cli", "-r", "reads.fq", "-a", "asm.fa", "-k", "37", "-t", "16", "-p", "2"], cwd=d, env=env, capture_output=True, text=True)
print("wall under cProfile %.2f s rc %d" % (time.perf_counter() - t0, p.returncode))
print("".join(ln + "\n" for ln in p.stderr.splitlines() if ln.startswith("[timing]")))
st = pstats.Stats(os.path.join(d, "prof.out")); st.sort_stats("tottime").print_stats(22)
shutil.rmtree(d, ignore_errors=True)
