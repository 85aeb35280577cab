#!/usr/bin/env python3
"""GPU box: what bounds files -> table (src/jasper.sh:177 `zcat -f $READS | jellyfish count`)?  The stages of the chunk loop of
jasper_count_reads_files timed one on top of the other on the same stream of text (JASPER_INGEST_STAGE in ingest_gpu.hip):

    1  page cache -> pinned buffer (READ_THREADS threads copy out of a mapping of the file; JASPER_INGEST_MMAP=0: pread)   for 4 / 8 / 16 / 32 reader threads
    2  + the copy to the device
    3  + the parsing kernels (newline numbering, FASTQ check, base compaction; the next chunk's copy runs beside them: JASPER_INGEST_OVERLAP=0 = one after the other)
    0  + counting (everything)

The stream: the configs[1] read file (2.9 GB of FASTQ) listed REPEAT times (default 8 = 23 GB, the size of one rank's share of
configs[3]); the page cache holds it once, which is what a warm cache gives a real 23-GB set as well -- far beyond any CPU cache.
    python tools/probes/ingest_stages.py [REPEAT]"""
import os, shutil, sys, tempfile, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from jasper_amd import synth
repeat = int(sys.argv[1]) if len(sys.argv) > 1 else 8
d = tempfile.mkdtemp(prefix="jasper_ingest_", dir=os.environ.get("TMPDIR", "/tmp"))
try:
    t0 = time.perf_counter()
    synth.write_cli_inputs(d, 47, 2, coverage=30)
    fq = os.path.join(d, "reads.fq")
    size = os.path.getsize(fq)
    print("wrote %.2f GB in %.1f s; cpus %s, affinity %d" % (size / 1e9, time.perf_counter() - t0, os.cpu_count(), len(os.sched_getaffinity(0))), flush=True)
    try:
        print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
    except OSError:
        pass
    from jasper_amd import KmerTable
    paths = [fq] * repeat
    t = KmerTable(37, min_slots=1 << 29)
    t.count_files([fq])          # buffers, code objects, the table's first touch
    t.sync()
    for stage, what in ((1, "page cache -> pinned"), (2, "+ H2D"), (3, "+ parse kernels"), (0, "+ counting (everything)")):
        for threads in ((4, 8, 16, 32) if stage == 1 else (8, 16, 32) if stage in (2, 3) else (16,)):
            os.environ["JASPER_INGEST_STAGE"] = str(stage)
            os.environ["JASPER_INGEST_READ_THREADS"] = str(threads)
            best = None
            for rep in range(2):
                t.clear()
                t.sync()
                t0 = time.perf_counter()
                t.count_files(paths)
                t.sync()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            print("stage %d %-26s %2d reader threads: %.3f s = %5.1f GB/s of text" % (stage, what, threads, best, size * repeat / best / 1e9), flush=True)
    os.environ.pop("JASPER_INGEST_STAGE", None)
finally:
    shutil.rmtree(d, ignore_errors=True)
