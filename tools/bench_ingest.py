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
