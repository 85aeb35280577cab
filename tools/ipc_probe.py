#!/usr/bin/env python3
"""does mapping the peers' slot arrays (hipIpcOpenMemHandle through jasper_table_attach_ipc) work for WORLD processes and
tables of 2^LOG2 slots?  Started under torch.distributed.run; every rank makes a table, the ranks exchange handles over gloo
and attach one at a time, then look a key up through the sharded view.   python tools/ipc_probe.py LOG2_SLOTS [one_gpu]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from jasper_amd import KmerTable

log2 = int(sys.argv[1])
one_gpu = len(sys.argv) > 2 and sys.argv[2] == "one_gpu"
rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
devi = 0 if one_gpu else local
torch.cuda.set_device(devi)
dev = torch.device("cuda", devi)
dist.init_process_group("gloo", rank=rank, world_size=world)
t = KmerTable(37, min_slots=1 << log2, device=devi)
t.count_bases(b"ACGTTGCATGCAAGTCCGATAGGCTAACGTTTGACCATGACAGATTACAGGCATCGATCGGATC")
t.sync()
mine = torch.frombuffer(bytearray(t.ipc_handle()), dtype=torch.uint8).to(dev)
parts = [torch.empty_like(mine) for _ in range(world)]
dist.all_gather(parts, mine)
handles = [bytes(p.cpu().numpy().tobytes()) for p in parts]
t0 = time.time()
for turn in range(world):
    if turn == rank:
        t.attach_ipc(handles, rank)
        print("rank %d: attached %d peers of 2^%d slots in %.2f s" % (rank, world - 1, log2, time.time() - t0), flush=True)
    dist.barrier()
got = t.lookup(["ACGTTGCATGCAAGTCCGATAGGCTAACGTTTGACCA"])
print("rank %d: lookup through the sharded view -> %s" % (rank, got), flush=True)
dist.barrier()
t.close()
dist.barrier()
dist.destroy_process_group()
