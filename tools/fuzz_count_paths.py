#!/usr/bin/env python3
"""GPU box: random inputs through both ways a table is filled -- the partition passes (8- or 16-byte records, narrow or wide
tables, one or two list levels, one or several pieces, empty or filled table) and the direct kernel -- which must build the
same table (distinct keys, occurrences, histogram, sampled lookups).
   python tools/fuzz_count_paths.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from jasper_amd import KmerTable, synth

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
bad = 0
t_start = time.time()
for case in range(cases):
    rng = np.random.default_rng(seed0 * 100003 + case)
    k = int(rng.choice([int(rng.integers(8, 38)), 37, int(rng.integers(38, 65)), int(rng.integers(38, 65)), 41, 45, 64]))
    G = int(rng.choice([120_000, 400_000, 1_500_000, 3_000_000]))
    cov = int(rng.choice([30, 60, 120])) if G < 1_000_000 else int(rng.choice([20, 30]))
    rl = int(rng.choice([80, 100, 150, 151, 250]))
    err = float(rng.choice([0.0, 0.002, 0.01]))
    gen = torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30)))
    genome = synth.torch_genome(gen, G, dev)
    nreads = max(1, G * cov // rl)
    reads = synth.torch_reads_stream(gen, genome, nreads, rl, err)
    n = reads.numel()
    # oddities: runs of one base, of N, a low-complexity stretch, lower case
    for _ in range(int(rng.integers(0, 4))):
        a = int(rng.integers(0, max(1, n - 70000))); ln = int(rng.integers(10, 60000))
        kind = int(rng.integers(0, 4))
        if kind == 0: reads[a:a + ln] = ord("ACGT"[int(rng.integers(4))])
        elif kind == 1: reads[a:a + ln] = ord("N")
        elif kind == 2: reads[a:a + ln] = torch.tensor(list(b"ACAG" * (ln // 4 + 1))[:min(ln, n - a)], dtype=torch.uint8, device=dev)
        else: reads[a:a + ln] = reads[a:a + ln] | 0x20
    torch.cuda.synchronize()
    log2 = int(rng.choice([0, 0, 22, 24, 26, 28]))
    slots = max(int(1.25 * n * 2.1 / 10), 1 << log2)
    two_calls = bool(rng.integers(2))
    try:
        tp = KmerTable(k, min_slots=slots)
        if two_calls:
            h = (n // 2) // 16 * 16
            cut = int(torch.nonzero(reads[h:h + 4096] == ord("N"))[0]) + h + 1 if (reads[h:h + 4096] == ord("N")).any() else n   # a record boundary
            tp.count_bases_device(reads.data_ptr(), cut)
            if cut < n: tp.count_bases_device(reads.data_ptr() + cut, n - cut)
        else:
            tp.count_bases_device(reads.data_ptr(), n)
        path, parts = tp.count_path(), tp.count_stages()[1]
        os.environ["JASPER_COUNT_DIRECT"] = "1"
        td = KmerTable(k, min_slots=slots)
        td.count_bases_device(reads.data_ptr(), n)
        del os.environ["JASPER_COUNT_DIRECT"]
        g = genome[:50_000].cpu().numpy().tobytes().decode()
        qs = [g[i:i + k] for i in range(0, len(g) - k, 613)] + ["A" * k, "ACAG" * 16, "N" * k]
        ok = tp.info()["distinct"] == td.info()["distinct"] and tp.info()["occurrences"] == td.info()["occurrences"] and tp.histogram() == td.histogram() and tp.lookup(qs) == td.lookup(qs)
        print("case %3d: k %2d G %7d cov %3d rl %3d err %.3f slots 2^%d%s: partitioned pieces %d %s" % (
            case, k, G, cov, rl, err, tp.info()["slots"].bit_length() - 1, " two calls" if two_calls else "", parts, "ok" if ok else "DIFFERENT"), flush=True)
        if not ok:
            bad += 1
            print("   ", tp.info(), td.info())
        tp.close(); td.close()
    except Exception as e:      # noqa: BLE001
        bad += 1
        print("case %d: k %d: EXCEPTION %r" % (case, k, e), flush=True)
        os.environ.pop("JASPER_COUNT_DIRECT", None)
    del reads, genome
    torch.cuda.empty_cache()
print("%d cases, %d failing, %.0f s" % (cases, bad, time.time() - t_start))
sys.exit(1 if bad else 0)
