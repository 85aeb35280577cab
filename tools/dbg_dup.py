#!/usr/bin/env python3
"""debug: path-1 table vs direct table for the input of test_atomic_free_counting_paths_equal_direct_counting (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jasper_amd import KmerTable, synth
k = int(sys.argv[1]) if len(sys.argv) > 1 else 37
G = 1_500_000
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(100 + k)
genome = synth.torch_genome(gen, G, dev)
nreads = G * 30 // 150
reads = synth.torch_reads_stream(gen, genome, nreads, 150, 0.004)
reads[1000:1600] = ord("A")
reads[5000:5400] = torch.tensor(list(b"ACACACACACACACACACAC" * 20), dtype=torch.uint8, device=dev)
reads[9000:9300] = ord("N")
torch.cuda.synchronize()
slots = int(1.25 * nreads * 150 * 2.1 / 10)
os.environ["JASPER_COUNT_DEBUG"] = "2"
tp = KmerTable(k, min_slots=slots)
tp.count_bases_device(reads.data_ptr(), reads.numel())
print("path", tp.count_path(), tp.info(), flush=True)
os.environ["JASPER_COUNT_DIRECT"] = "1"
td = KmerTable(k, min_slots=slots)
td.count_bases_device(reads.data_ptr(), reads.numel())
del os.environ["JASPER_COUNT_DIRECT"]
print("direct", td.info(), flush=True)
ep = np.array(tp.export_entries()).reshape(-1, 3)
ed = np.array(td.export_entries()).reshape(-1, 3)
print("entries", ep.shape, ed.shape)
def keyed(e):
    o = np.lexsort((e[:, 1], e[:, 0]))
    return e[o]
ep, ed = keyed(ep), keyed(ed)
same = (ep[1:, 0] == ep[:-1, 0]) & (ep[1:, 1] == ep[:-1, 1])
print("duplicate keys in the path-1 table:", int(same.sum()))
s = tp.info()["slots"].bit_length() - 1
B = 2 * k
idx = np.nonzero(same)[0][:20]
for i in idx:
    hi, lo = int(ep[i, 0]), int(ep[i, 1])
    h = (hi << 64) | lo
    home = h >> (B - s)
    print("dup hash %x home %d (local %d of region %d) counts %d + %d" % (h, home, home & 4095, home >> 12, int(ep[i, 2]), int(ep[i + 1, 2])))
# keys of one table that the other lacks
sp = set(map(tuple, ep[:, :2].tolist())); sd = set(map(tuple, ed[:, :2].tolist()))
print("only in path-1:", len(sp - sd), "only in direct:", len(sd - sp))
