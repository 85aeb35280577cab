"""First contact with the transport, in throw-away processes: run as one rank per GPU under torch.distributed.run, BEFORE a
multi-GPU job commits to the backend (bench.py --gpus N, role of nothing in the reference: src/jasper.sh is one node, one process
tree).  A collective that never returns would hang the job; here it costs a killed self-test and a fall-back.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        -m jasper_amd._selftest --out FILE [--backend nccl|gloo] [--one-gpu] [--mb 64]

Every rank: process group, ONE all_to_all_single of --mb megabytes of int64 (the call that moves the region lists, checked word
by word), one all_reduce, one all_gather, a barrier; then the pattern of the pipelined exchange (an asynchronous all_to_all with
agreements on the control group beside it: "pipeline_ok").  Rank 0 writes {"ok", "world", "backend", "ms": {...}, "error"} to FILE.
"""
import argparse
import datetime
import json
import os
import sys
import time


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--one-gpu", action="store_true")
    ap.add_argument("--mb", type=int, default=64)
    a = ap.parse_args()
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    res = {"ok": False, "world": world, "backend": a.backend, "ms": {}, "error": None}
    import torch
    import torch.distributed as dist
    try:
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible")
        di = 0 if a.one_gpu else local
        torch.cuda.set_device(di)
        dev = torch.device("cuda", di)
        t0 = time.perf_counter()
        tmo = datetime.timedelta(seconds=60)
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world, timeout=tmo)
        res["ms"]["init"] = round((time.perf_counter() - t0) * 1e3, 1)
        per = max(1, (a.mb << 20) // 8 // world)                     # words per destination
        send = (torch.arange(world * per, dtype=torch.int64, device=dev) % per) + (rank * world + torch.arange(world, device=dev).repeat_interleave(per)) * (1 << 32)
        recv = torch.empty_like(send)
        torch.cuda.synchronize(dev)
        for name in ("all_to_all_single_first", "all_to_all_single"):
            t0 = time.perf_counter()
            if a.backend == "nccl":
                dist.all_to_all_single(recv, send)
            else:                                                     # (gloo has no all_to_all for device tensors: the rehearsal path of dist._all_to_all_rows)
                parts = [torch.empty_like(send) for _ in range(world)]
                dist.all_gather(parts, send)
                recv = torch.cat([p.view(world, per)[rank] for p in parts])
            torch.cuda.synchronize(dev)
            res["ms"][name] = round((time.perf_counter() - t0) * 1e3, 2)
        want = (torch.arange(world * per, dtype=torch.int64, device=dev) % per) + (torch.arange(world, device=dev).repeat_interleave(per) * world + rank) * (1 << 32)
        if not bool((recv == want).all().item()):
            raise RuntimeError("all_to_all_single delivered wrong words")
        t0 = time.perf_counter()
        one = torch.tensor([rank + 1], dtype=torch.int64, device=dev)
        dist.all_reduce(one)
        if int(one.item()) != world * (world + 1) // 2:
            raise RuntimeError("all_reduce gave a wrong sum")
        g = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(g, torch.tensor([rank], dtype=torch.int64, device=dev))
        if [int(x.item()) for x in g] != list(range(world)):
            raise RuntimeError("all_gather gave wrong values")
        dist.barrier()
        torch.cuda.synchronize(dev)
        res["ms"]["all_reduce_all_gather_barrier"] = round((time.perf_counter() - t0) * 1e3, 2)
        res["bytes_all_to_all"] = int(send.numel() * 8)
        res["ok"] = True
        # the pattern of the pipelined exchange (dist.count_sharded): a list all_to_all under way, not waited for, while small
        # agreements go through the control group on host tensors -- its own verdict (a failure here costs the pipeline, not RCCL)
        res["pipeline_ok"] = False
        try:
            from jasper_amd import dist as jdist
            t0 = time.perf_counter()
            ctrl = jdist._control_group(None)
            res["ms"]["control_group"] = round((time.perf_counter() - t0) * 1e3, 1)
            t0 = time.perf_counter()
            get, wait = jdist._all_to_all_rows_async(send.view(world, per), None)
            cdev = torch.device("cpu") if dist.get_backend(ctrl) == "gloo" else dev
            for i in range(3):
                c = torch.tensor([rank + 1 + i], dtype=torch.int64, device=cdev)
                dist.all_reduce(c, group=ctrl)
                if int(c.item()) != world * (world + 1) // 2 + world * i:
                    raise RuntimeError("control all_reduce gave a wrong sum")
            wait()
            if not bool((get().reshape(-1) == want).all().item()):
                raise RuntimeError("the asynchronous all_to_all delivered wrong words")
            res["ms"]["async_all_to_all_with_agreements"] = round((time.perf_counter() - t0) * 1e3, 2)
            res["pipeline_ok"] = True
        except Exception as e:      # noqa: BLE001
            res["pipeline_error"] = "rank %d: %r" % (rank, e)
    except Exception as e:      # noqa: BLE001 -- the verdict is the file
        res["error"] = "rank %d: %r" % (rank, e)
    # every rank's verdict counts: a rank that failed writes its own file next to rank 0's
    try:
        with open(a.out + (".rank%d" % rank if rank else ""), "w") as f:
            json.dump(res, f)
    finally:
        try:
            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:       # noqa: BLE001
            pass
    sys.exit(0 if res["ok"] else 3)


if __name__ == "__main__":
    main()
