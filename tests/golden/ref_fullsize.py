#!/usr/bin/env python3
"""Build container only: the REAL reference (bash src/jasper.sh + Jellyfish 2.3.0 + unmodified jasper.py, 8 vCPU) on the
full-size cfg-2 synthetic input (47 Mb genome, 30x 150-bp reads, k=37, 2 passes) -> tests/golden/fullsize_cfg2.json:
wall time per stage and digests of its outputs.  tests/test_gpu_cli_fullsize.py regenerates the same input on the GPU
box (jasper_amd.synth.write_cli_inputs is deterministic), runs `python -m jasper_amd.cli` and compares the digests.

usage: python3 tests/golden/ref_fullsize.py [genome_mb] [threads] [k] [passes] [seed] [name] [coverage] [contigs] [populations]
       (defaults = cfg 2: 47 8 37 2 2 fullsize_cfg2 30 1 1;   cfg 1 = 4.6 8 25 1 1 fullsize_cfg1)
       the shapes of SURVEY 8d (what tests/test_gpu_cli_fullsize.py runs):
         cfg 2 chunked as -t 16      47   16 37 2 2 fullsize_cfg2_t16
         cfg 3 (7 contigs, 40x)      140  16 37 2 3 fullsize_cfg3 40 7
         cfg 3 at 1/4 scale          35   16 37 2 3 fullsize_cfg3_quarter 40 7
         cfg 4 shape, 1/64 scale     48.4 64 37 2 4 fullsize_cfg4_scaled 30 24   (24 contigs, several chunks each, many batch files)
         cfg 5 shape, scaled         4    16 37 4 5 fullsize_cfg5_scaled 30 3 10 (10 read sets with private SNPs, 4 passes)
         the same, too thin          10   16 37 4 5 fullsize_cfg5_lowcov 10 3 10 (first local minimum < 4: the reference aborts)"""
import json, os, shutil, subprocess, sys, tempfile, time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_golden as G
from jasper_amd import synth

gmb = float(sys.argv[1]) if len(sys.argv) > 1 else 47.0
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
K = int(sys.argv[3]) if len(sys.argv) > 3 else 37
P = int(sys.argv[4]) if len(sys.argv) > 4 else 2
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 2
name = (sys.argv[6] if len(sys.argv) > 6 else "fullsize_cfg2") + ".json"
coverage = int(sys.argv[7]) if len(sys.argv) > 7 else 30
contigs = int(sys.argv[8]) if len(sys.argv) > 8 else 1
populations = int(sys.argv[9]) if len(sys.argv) > 9 else 1
work = tempfile.mkdtemp(prefix="ref_full_", dir="/tmp")
run_dir = os.path.join(work, "run")
os.makedirs(run_dir)
t0 = time.time()
nreads, asm_len = synth.write_cli_inputs(run_dir, gmb, seed, coverage=coverage, contigs=contigs, populations=populations)
print("inputs: %d reads, %d assembly bases, %.1f s" % (nreads, asm_len, time.time() - t0), flush=True)
pp = os.path.join(work, "pp")
os.makedirs(os.path.join(pp, "Bio"))
for fn in ("dna_jellyfish.py", "_dna_jellyfish.so"):
    shutil.copy(os.path.join(G.JF_PY, fn), pp)
shutil.copy(os.path.join(G.REF, "jellyfish.py"), pp)
open(os.path.join(pp, "Bio", "__init__.py"), "w").write("")
stub = G.DRIVER.split("# --- stand-in")[1].split("Bio = types.ModuleType")[0]
open(os.path.join(pp, "Bio", "pairwise2.py"), "w").write(
    "# stand-in" + stub + "\nimport types\nalign = types.SimpleNamespace(globalms=_globalms)\ndef format_alignment(*a, **k): return ''\n")
bindir = os.path.join(work, "bin")
os.makedirs(bindir)
for fn in ("jasper.sh", "jasper.py", "jellyfish.py"):
    shutil.copy(os.path.join(G.REF, fn), bindir)
    os.chmod(os.path.join(bindir, fn), 0o755)
# The reference removes its {0,P}qValCalcHelper.csv files at the end (src/jasper.sh:258), and without `bc` in this container its
# log says "Q value = Inf": an `rm` of our own, first on PATH, keeps a copy of those files (the per-batch "bad total" lines
# src/jasper.py:107-111 appended) before it removes them, so that the QV INPUTS of the real run are pinned.
keep = os.path.join(work, "kept")
os.makedirs(keep)
open(os.path.join(bindir, "rm"), "w").write(
    "#!/bin/bash\nfor a in \"$@\"; do case \"$a\" in *qValCalcHelper.csv) [ -f \"$a\" ] && cp \"$a\" %s/ ;; esac; done\nexec /bin/rm \"$@\"\n" % keep)
os.chmod(os.path.join(bindir, "rm"), 0o755)
env = dict(os.environ, PATH=bindir + ":" + os.path.dirname(G.JF_BIN) + ":" + os.environ["PATH"], PYTHONPATH=pp,
           LD_LIBRARY_PATH=os.path.join(os.path.dirname(os.path.dirname(G.JF_BIN)), "lib"))
t1 = time.time()
p = subprocess.run(["bash", os.path.join(bindir, "jasper.sh"), "-r", " ".join(synth.read_files(populations)), "-a", "asm.fa", "-k", str(K), "-t", str(threads), "-p", str(P)],
                   cwd=run_dir, env=env, capture_output=True, text=True)
wall = time.time() - t1
print(p.stdout[-3000:])
print(p.stderr[-1500:])
print("exit", p.returncode, "wall %.1f s" % wall, flush=True)
out = dict(genome_mb=gmb, coverage=coverage, contigs=contigs, populations=populations, threads=threads, k=K, passes=P, seed=seed, reads=nreads, assembly_bases=asm_len, exit=p.returncode,
           reference_wall_seconds=round(wall, 1), host="build container, %d vCPU" % (os.cpu_count() or 0),
           stdout=[G.re_sub_date(l) for l in p.stdout.splitlines()])
if p.returncode == 0:
    out.update(synth.output_digests(run_dir, k=K))
    for key, fn in (("qv_before", "0qValCalcHelper.csv"), ("qv_after", "%dqValCalcHelper.csv" % P)):
        rows = [l.split() for l in open(os.path.join(keep, fn)).read().splitlines() if l.strip()]
        out[key] = [sum(int(r[0]) for r in rows), sum(int(r[1]) for r in rows)]      # exact integer column sums (gawk's behaviour)
        out[key + "_lines"] = len(rows)
json.dump(out, open(os.path.join(HERE, name), "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in out.items() if k != "stdout"}, indent=1))
shutil.rmtree(work)
