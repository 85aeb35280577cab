import os, shutil, sys, tempfile, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT)
from jasper_amd import synth, KmerTable
d = tempfile.mkdtemp(prefix="jasper_ingest_", dir="/tmp")
try:
    synth.write_cli_inputs(d, 47, 2, coverage=30)
    fq = os.path.join(d, "reads.fq"); size = os.path.getsize(fq)
    paths = [fq] * 8
    t = KmerTable(37, min_slots=1 << 29)
    t.count_files([fq]); t.sync()
    for stage, what in ((1, "page cache -> pinned"), (2, "+ H2D"), (3, "+ parse kernels"), (0, "+ counting (everything)")):
        for threads in (8, 16):
            for mm, ov in ((0, 0), (1, 0), (1, 1), (0, 0), (1, 0), (1, 1)):
                if stage in (1, 2) and ov: continue          # (the probe's early stages run the chunk loop without the overlap)
                os.environ["JASPER_INGEST_STAGE"] = str(stage); os.environ["JASPER_INGEST_READ_THREADS"] = str(threads); os.environ["JASPER_INGEST_MMAP"] = str(mm)
                os.environ["JASPER_INGEST_OVERLAP"] = str(ov)
                t.clear(); t.sync(); t0 = time.perf_counter(); t.count_files(paths); t.sync(); dt = time.perf_counter() - t0
                info = t.info()
                print("stage %d %-26s %2d threads mmap %d overlap %d: %.3f s = %5.1f GB/s   (distinct %d, occurrences %d)" % (stage, what, threads, mm, ov, dt, size * 8 / dt / 1e9, info["distinct"], info["occurrences"]), flush=True)
    info = t.info(); print(info["distinct"], info["occurrences"])
finally:
    shutil.rmtree(d, ignore_errors=True)
