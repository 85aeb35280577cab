#!/usr/bin/env python3
"""PCIe-inclusive rates: host base stream -> table (jasper_count_bases), FASTQ file -> table (jasper_count_reads_files)"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from jasper_amd import KmerTable, synth
rng = np.random.default_rng(1)
G = 8_000_000
genome = synth.make_genome(rng, G)
reads = synth.make_reads_stream(rng, genome, 30, 150, 0.003)
b = reads.tobytes()
kmers = (len(b) // 151) * (150 - 37 + 1)
t = KmerTable(37, min_slots=1 << 27)
t0 = time.perf_counter(); t.count_bases(b); t.sync(); t1 = time.perf_counter()
print("host base stream: %.0f MB in %.1f ms -> %.2f GB/s, %.2f Gk-mers/s (PCIe-inclusive)" % (len(b) / 1e6, (t1 - t0) * 1e3, len(b) / (t1 - t0) / 1e9, kmers / (t1 - t0) / 1e9), flush=True)
t.clear()
# FASTQ file
d = tempfile.mkdtemp()
fq = os.path.join(d, "r.fq")
r2 = reads.reshape(-1, 151)[:, :150]
with open(fq, "wb") as f:
    q = b"I" * 150
    for i in range(r2.shape[0]):
        f.write(b"@r%d\n" % i); f.write(r2[i].tobytes()); f.write(b"\n+\n"); f.write(q); f.write(b"\n")
sz = os.path.getsize(fq)
for label, env in (("GPU text parser", None), ("GPU text parser (again)", None), ("host state machine", "1")):
    if env:
        os.environ["JASPER_INGEST_HOST"] = env
    t.clear()
    t0 = time.perf_counter(); t.count_files([fq]); t.sync(); t1 = time.perf_counter()
    print("FASTQ file, %s: %.0f MB in %.1f ms -> %.2f GB/s of text, %.2f Gk-mers/s (file read + PCIe inclusive); parsed on GPU/host: %s"
          % (label, sz / 1e6, (t1 - t0) * 1e3, sz / (t1 - t0) / 1e9, kmers / (t1 - t0) / 1e9, t.last_ingest()), flush=True)
os.environ.pop("JASPER_INGEST_HOST", None)
# gzip input: the text split into 4 .gz files (random qualities, so that it compresses like real FASTQ, ~3.5x)
import gzip, subprocess
qs = np.frombuffer(b"FFFFFFFF:,#", dtype=np.uint8)
parts = []
per = r2.shape[0] // 4
for j in range(4):
    p = os.path.join(d, "p%d.fq" % j)
    rows = r2[j * per:(j + 1) * per]
    rec = np.empty((rows.shape[0], 4 + 150 + 3 + 150 + 1), dtype=np.uint8)
    rec[:, 0:4] = np.frombuffer(b"@rd\n", dtype=np.uint8)
    rec[:, 4:154] = rows
    rec[:, 154:157] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, 157:307] = rng.choice(qs, (rows.shape[0], 150))
    rec[:, 307] = 10
    rec.tofile(p)
    subprocess.run(["gzip", "-1", "-f", p], check=True)
    parts.append(p + ".gz")
gsz = sum(os.path.getsize(p) for p in parts)
text = 4 * per * 308
for label, mb in (("one file at a time (JASPER_INGEST_AHEAD_MB=0: 8 MB look-ahead)", "0"), ("all files inflated ahead (default budget)", None)):
    if mb is None:
        os.environ.pop("JASPER_INGEST_AHEAD_MB", None)
    else:
        os.environ["JASPER_INGEST_AHEAD_MB"] = mb
    t.clear()
    t0 = time.perf_counter(); t.count_files(parts); t.sync(); t1 = time.perf_counter()
    print("4 gzip files (%.0f MB -> %.0f MB of text), %s: %.2f s -> %.2f GB/s of text" % (gsz / 1e6, text / 1e6, label, t1 - t0, text / (t1 - t0) / 1e9), flush=True)
# ONE large gzip file (what a real read set looks like: R1.fastq.gz, R2.fastq.gz): inflated by many threads (pgunzip.hpp)
big = os.path.join(d, "big.fq")
with open(big, "wb") as f:
    for rep in range(3):
        for p in parts:
            f.write(gzip.open(p).read())
subprocess.run(["gzip", "-6", "-f", big], check=True)
big += ".gz"
bsz = os.path.getsize(big)
btext = 3 * text
import ctypes as C
from jasper_amd import _lib
L = _lib.lib()
for thr in (1, 8, 16, 32, 48, 64):
    n = C.c_uint64(0); par = C.c_int(0)
    t0 = time.perf_counter()
    rc = L.jasper_inflate_file(big.encode(), thr, 0, None, C.byref(n), C.byref(par))
    t1 = time.perf_counter()
    print("one gzip file (%.0f MB -> %.0f MB of text), inflate only, %2d threads (%s): %.2f s -> %.2f GB/s of text"
          % (bsz / 1e6, n.value / 1e6, thr, "parallel" if par.value else "zlib", t1 - t0, n.value / (t1 - t0) / 1e9), flush=True)
for thr in ("1", None):
    if thr:
        os.environ["JASPER_INGEST_GZ_THREADS"] = thr
    else:
        os.environ.pop("JASPER_INGEST_GZ_THREADS", None)
    t.clear()
    t0 = time.perf_counter(); t.count_files([big]); t.sync(); t1 = time.perf_counter()
    print("one gzip file -> table, %s: %.2f s -> %.2f GB/s of text, %.2f Gk-mers/s"
          % ("zlib reader (JASPER_INGEST_GZ_THREADS=1)" if thr else "default threads", t1 - t0, btext / (t1 - t0) / 1e9, 3 * 4 * per * 114 / (t1 - t0) / 1e9), flush=True)
