#!/usr/bin/env python3
"""GPU box: the read feed on a file that is / is not in the page cache (evicted with posix_fadvise DONTNEED after an fsync): which reader
it picks (JASPER_COUNT_DEBUG line) and that the table is the same."""
import os, shutil, sys, tempfile, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT)
os.environ["JASPER_COUNT_DEBUG"] = "1"
from jasper_amd import synth, KmerTable
d = tempfile.mkdtemp(prefix="jasper_cold_", dir=os.environ.get("TMPDIR", "/tmp"))
try:
    synth.write_cli_inputs(d, 8, 3, coverage=30)
    fq = os.path.join(d, "reads.fq"); size = os.path.getsize(fq)
    res = []
    for evict in (False, True, False):
        if evict:
            fd = os.open(fq, os.O_RDONLY); os.fsync(fd); os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED); os.close(fd)
        t = KmerTable(37, min_slots=1 << 27)
        a = time.perf_counter(); t.count_files([fq]); t.sync(); b = time.perf_counter()
        i = t.info(); res.append((i["distinct"], i["occurrences"]))
        print("evicted first: %s -> %.3f s for %.2f GB, distinct %d" % (evict, b - a, size / 1e9, i["distinct"]), flush=True)
        t.close()
    assert res[0] == res[1] == res[2]
    print("same table every time")
finally:
    shutil.rmtree(d, ignore_errors=True)
