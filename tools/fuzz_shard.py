#!/usr/bin/env python3
"""the GPU fuzz (HIP path vs oracle, tests/test_gpu_fuzz.py) with every READ going through an owner-sharded view: the table
is counted as usual, then split into 2..5 owner tables (export grouped by owner -> LDS-region import), and histogram,
lookups and polishing run through the shards.   python tools/fuzz_shard.py SEED0 N [logfile]"""
import os, sys, tempfile, pathlib, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def sharded_class(KmerTable):
    import torch

    class Sharded(KmerTable):
        def _split(self):
            if getattr(self, "_shards", None) is not None:
                return self._shards
            base = KmerTable.info(self)
            n = 2 + base["distinct"] % 4
            cap = base["distinct"] // n + base["distinct"] // 8 + 4096
            while True:
                buf = torch.empty((n, cap, 2), dtype=torch.int64, device="cuda")
                torch.cuda.synchronize()
                counts = KmerTable.export_owner(self, buf.data_ptr(), cap, n)
                if max(counts) <= cap:
                    break
                cap = max(counts)
            assert sum(counts) == base["distinct"]
            shards = []
            for o in range(n):
                t = KmerTable(self.k, min_slots=base["slots"])
                assert KmerTable.info(t)["slots"] == base["slots"]
                if counts[o]:
                    t.import_packed_multi([buf[o].data_ptr()], [counts[o]])
                shards.append(t)
            for o, t in enumerate(shards):
                t.attach_tables(shards, o)
            self._shards = shards
            return shards

        def info(self):
            d = KmerTable.info(self)
            d["distinct"] = sum(KmerTable.info(t)["distinct"] for t in self._split())
            return d

        def histogram(self):
            acc = [0] * 10002
            for t in self._split():
                acc = [a + b for a, b in zip(acc, KmerTable.histogram(t))]
            return acc

        def lookup(self, strings):
            return KmerTable.lookup(self._split()[0], strings)

        def polish_batch(self, seqs, thre, passes, fix=True):
            return KmerTable.polish_batch(self._split()[-1], seqs, thre, passes, fix=fix)

        def close(self):
            for t in getattr(self, "_shards", None) or []:
                t.close()
            self._shards = None
            KmerTable.close(self)

    return Sharded


if __name__ == "__main__":
    import test_gpu_fuzz as T
    from jasper_amd import KmerTable, polisher
    from oracle import oracle as O
    import make_golden as G
    import fuzz_vs_reference as F
    seed0, n = int(sys.argv[1]), int(sys.argv[2])
    log = open(sys.argv[3], "a") if len(sys.argv) > 3 else sys.stdout
    S = sharded_class(KmerTable)
    tmp = pathlib.Path(tempfile.mkdtemp(prefix="fuzzshard_"))
    t0 = time.time()
    bad = 0
    for i, seed in enumerate(range(seed0, seed0 + n)):
        try:
            T._one(seed, S, polisher, O, G, F, tmp)
        except AssertionError as e:
            bad += 1
            log.write("seed %d FAILED: %s\n" % (seed, str(e)[:300]))
        for f in tmp.iterdir():
            f.unlink()
        if (i + 1) % 100 == 0:
            log.write("%d cases, %d failing, %.0f s\n" % (i + 1, bad, time.time() - t0))
            log.flush()
    log.write("done (sharded reads): seeds %d..%d, %d failing\n" % (seed0, seed0 + n - 1, bad))
