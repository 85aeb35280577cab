import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from jasper_amd import KmerTable, synth
dev = torch.device("cuda", 0)
G = 6_000_000
g = synth.torch_genome(torch.Generator(device=dev).manual_seed(1), G, dev)
n = G * 30 // 150
ra = synth.torch_reads_stream(torch.Generator(device=dev).manual_seed(2), g, n // 2)
torch.cuda.synchronize()
ms = 1 << 27
P = KmerTable(37, ms); P.count_bases_device(ra.data_ptr(), ra.numel())
os.environ["JASPER_COUNT_DIRECT"] = "1"
D = KmerTable(37, ms); D.count_bases_device(ra.data_ptr(), ra.numel())
del os.environ["JASPER_COUNT_DIRECT"]
print("partitioned", P.info(), P.count_stages()); print("direct", D.info(), D.count_stages())
print("hist equal", P.histogram() == D.histogram())
def entries(t):
    n = t.export_packed(0, 0)
    b = torch.zeros((n, 2), dtype=torch.int64, device=dev); t.export_packed(b.data_ptr(), n)
    x = b.cpu().numpy().view(np.uint64)
    return x
ep, ed = entries(P), entries(D)
kp = ep[:, 0].astype(object) | ((ep[:, 1] & np.uint64(1023)).astype(object) << 64)
print("dup keys in partitioned:", len(kp) - len(set(kp.tolist())))
sp = set(map(tuple, ep.tolist())); sd = set(map(tuple, ed.tolist()))
print("entry sets equal (hash,count):", sp == sd, len(sp - sd), len(sd - sp))
# probe-invariant check by lookups of all genome windows
kd = ed[:, 0].astype(object) | ((ed[:, 1] & np.uint64(1023)).astype(object) << 64)
print("dup keys in direct:", len(kd) - len(set(kd.tolist())))
e3p = P.export_entries(); e3d = D.export_entries()
k3p = set(((int(a) << 64) | int(b), int(c)) for a, b, c in e3p.tolist()); k3d = set(((int(a) << 64) | int(b), int(c)) for a, b, c in e3d.tolist())
print("24-B export: n", len(e3p), len(e3d), "unique", len(k3p), len(k3d), "equal", k3p == k3d)
kp2 = set(((int(hi) & 1023) << 64 | int(lo), int(hi) >> 10) for lo, hi in ep.tolist())
print("packed vs 24B (partitioned):", kp2 == k3p, len(kp2), len(k3p), len(kp2 - k3p))
