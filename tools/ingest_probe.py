#!/usr/bin/env python3
"""GPU box: where the time of `files -> table` goes on THIS box (the stage varies 0.16 ... 1.7 s between boxes): raw read rate of the
FASTQ from the page cache, library load + first HIP call, table allocation, count_files"""
import os, sys, time, tempfile, shutil
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from jasper_amd import synth
d = tempfile.mkdtemp(prefix="jasper_ingest_", dir=os.environ.get("TMPDIR", "/tmp"))
t0 = time.perf_counter()
nreads, asm_len = synth.write_cli_inputs(d, 47, 2, coverage=30)
fq = os.path.join(d, "reads.fq")
print("wrote %.2f GB in %.1f s; MemAvailable %s" % (os.path.getsize(fq) / 1e9, time.perf_counter() - t0, [l for l in open("/proc/meminfo") if l.startswith(("MemAvailable", "Cached", "Dirty"))]), flush=True)
try:
    print("cgroup memory.max:", open("/sys/fs/cgroup/memory.max").read().strip(), "current:", open("/sys/fs/cgroup/memory.current").read().strip())
except OSError as e:
    print("no cgroup v2 files:", e)
for rep in range(2):
    t0 = time.perf_counter()
    n = 0
    with open(fq, "rb", buffering=0) as f:
        buf = bytearray(64 << 20)
        while True:
            k = f.readinto(buf)
            if not k:
                break
            n += k
    dt = time.perf_counter() - t0
    print("raw read %d: %.2f GB in %.3f s = %.1f GB/s" % (rep, n / 1e9, dt, n / dt / 1e9), flush=True)
t0 = time.perf_counter()
from jasper_amd import KmerTable
t = KmerTable(37, min_slots=int(1.25 * os.path.getsize(fq) / 10))
t.sync()
print("library + HIP start + table: %.3f s" % (time.perf_counter() - t0), flush=True)
for rep in range(3):
    t.clear()
    if rep == 0:
        os.environ["JASPER_COUNT_DEBUG"] = "1"
    else:
        os.environ.pop("JASPER_COUNT_DEBUG", None)
    t0 = time.perf_counter()
    t.count_files([fq])
    t.sync()
    dt = time.perf_counter() - t0
    print("count_files %d: %.3f s = %.1f GB/s of text; %s" % (rep, dt, os.path.getsize(fq) / dt / 1e9, t.last_ingest()), flush=True)
shutil.rmtree(d, ignore_errors=True)
