import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jasper_amd import KmerTable, synth
dev = torch.device("cuda", 0)
G = 6_000_000
g = synth.torch_genome(torch.Generator(device=dev).manual_seed(1), G, dev)
n = G * 30 // 150
ra = synth.torch_reads_stream(torch.Generator(device=dev).manual_seed(2), g, n // 2)
rb = synth.torch_reads_stream(torch.Generator(device=dev).manual_seed(3), g, n // 2)
torch.cuda.synchronize()
ms = 1 << 27
A, B, F = KmerTable(37, ms), KmerTable(37, ms), KmerTable(37, ms)
A.count_bases_device(ra.data_ptr(), ra.numel()); B.count_bases_device(rb.data_ptr(), rb.numel())
F.count_bases_device(ra.data_ptr(), ra.numel()); F.count_bases_device(rb.data_ptr(), rb.numel())
for name, t in (("A", A), ("B", B), ("F", F)):
    i = t.info()
    parts = [t.export_packed(0, 0, p, 2) for p in range(2)]
    print(name, i, "parts", parts, sum(parts), "all", t.export_packed(0, 0), "hist1-5", t.histogram()[1:6], flush=True)
# A owns part 0, B owns part 1
nb0 = B.export_packed(0, 0, 0, 2)
buf = torch.zeros((nb0, 2), dtype=torch.int64, device=dev)
print("export B part0", B.export_packed(buf.data_ptr(), nb0, 0, 2), nb0)
A.import_packed(buf.data_ptr(), nb0, 0)
print("A after add", A.info(), [A.export_packed(0, 0, p, 2) for p in range(2)], "F part0", F.export_packed(0, 0, 0, 2))
na0 = A.export_packed(0, 0, 0, 2)
out = torch.zeros((na0, 2), dtype=torch.int64, device=dev)
A.export_packed(out.data_ptr(), na0, 0, 2)
B.import_packed(out.data_ptr(), na0, 1)
print("B after set", B.info(), [B.export_packed(0, 0, p, 2) for p in range(2)])
import numpy as np
nb = B.export_packed(0, 0, 0, 2)
got = torch.zeros((nb, 2), dtype=torch.int64, device=dev)
B.export_packed(got.data_ptr(), nb, 0, 2)
o = out.cpu().numpy().view(np.uint64); gq = got.cpu().numpy().view(np.uint64)
mask10 = np.uint64((1 << 10) - 1)
def keyset(x):
    return set(zip(x[:, 0].tolist(), (x[:, 1] & mask10).tolist()))
ko, kg = keyset(o), keyset(gq)
missing = list(ko - kg)
print("missing", len(missing), "extra", len(kg - ko))
B_ = 74; s_ = 27
homes = sorted(((hi << 64) | lo) >> (B_ - s_) for lo, hi in missing[:200000])
import collections
print("first homes", homes[:10])
reg = collections.Counter(h >> 13 for h in homes)
print("regions hit", len(reg), "most common", reg.most_common(5))
loc = collections.Counter((h & 8191) >> 9 for h in homes)
print("position within 8192-slot region (in 512-slot bins)", sorted(loc.items()))
