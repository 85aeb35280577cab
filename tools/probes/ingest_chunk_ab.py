import os, shutil, sys, tempfile, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT)
from jasper_amd import synth, KmerTable
d = tempfile.mkdtemp(prefix="jasper_ingest_", dir="/tmp")
try:
    synth.write_cli_inputs(d, 47, 2, coverage=30)
    fq = os.path.join(d, "reads.fq"); size = os.path.getsize(fq)
    t0 = KmerTable(37, min_slots=1 << 29); t0.count_files([fq]); t0.sync(); t0.close()     # runtime, code objects
    for rep in range(2):
        for mib in (64, 32, 16, 8):
            os.environ["JASPER_INGEST_CHUNK"] = str(mib << 20)
            t = KmerTable(37, min_slots=1 << 29)
            a = time.perf_counter(); t.count_files([fq]); t.sync(); b = time.perf_counter()       # a fresh table: pinned + device buffers allocated in this call
            t.clear(); t.sync()
            c = time.perf_counter(); t.count_files([fq] * 8); t.sync(); e = time.perf_counter()
            print("chunk %2d MiB: first call on a fresh table (2.9 GB) %.3f s; 23 GB %.3f s = %.1f GB/s; distinct %d" % (mib, b - a, e - c, size * 8 / (e - c) / 1e9, t.info()["distinct"]), flush=True)
            t.close()
finally:
    shutil.rmtree(d, ignore_errors=True)
