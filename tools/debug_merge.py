"""debug: 2 ranks on one GPU (gloo), bench-like workload; compare merged histogram with a single-table count"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from jasper_amd import KmerTable, synth, dist as jd, polisher
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
G = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 6_000_000
genomes = [synth.torch_genome(torch.Generator(device=dev).manual_seed(2000 + j), G, dev) for j in range(world)]
whole = torch.cat(genomes)
nreads_total = G * world * 30 // 150
lo = rank * nreads_total // world; hi = (rank + 1) * nreads_total // world
reads = synth.torch_reads_stream(torch.Generator(device=dev).manual_seed(2500 + rank), whole, hi - lo)
torch.cuda.synchronize()
jf = int(nreads_total * 150 * 2.1 / 10)
t = KmerTable(37, min_slots=max(1 << 21, int(1.25 * jf)), device=0)
t.count_bases_device(reads.data_ptr(), reads.numel())
h0 = t.histogram(); i0 = t.info()
print(rank, "before merge", i0, h0[1:12], flush=True)
n = jd.merge_tables(t, dev)
h1 = t.histogram(); i1 = t.info()
print(rank, "after merge received", n, i1, h1[1:12], flush=True)
# reference: gather all reads on rank 0 and count in one table
sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
dist.all_gather(sizes, torch.tensor([reads.numel()]))
if rank == 0:
    t2 = KmerTable(37, min_slots=max(1 << 21, int(1.25 * jf)), device=0)
    t2.count_bases_device(reads.data_ptr(), reads.numel())
    other = synth.torch_reads_stream(torch.Generator(device=dev).manual_seed(2500 + 1), whole, (2 * nreads_total // world) - (nreads_total // world))
    torch.cuda.synchronize()
    t2.count_bases_device(other.data_ptr(), other.numel())
    h2 = t2.histogram()
    print("single table", t2.info(), h2[1:12], "equal:", h2 == h1, flush=True)
    print("threshold", polisher.threshold_from_histo_rows([(m, h1[m]) for m in range(1, 10002) if h1[m]]))
dist.barrier()
dist.destroy_process_group()
