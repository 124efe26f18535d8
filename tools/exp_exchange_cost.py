#!/usr/bin/env python3
"""Diagnostic: what the final exchange of the multi-rank benchmark costs on RCCL, step by step
(single GPU: a world of one rank)."""
import importlib
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("2d_object_detection_amd")
syn, util, dsb = pkg.synthetic, pkg.util, pkg.distributed
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29534")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
heads = [h.to(dev) for h in syn.make_heads(64, 416, 5, seed=1)]
engs = [util.YoloV3Post(64, 416, 5, 0.5, 0.4, device=dev) for _ in range(4)]
streams = [torch.cuda.Stream(device=dev) for _ in range(4)]


def T():
    torch.cuda.synchronize()
    return time.perf_counter()


for rnd in range(3):
    for i in range(40):
        engs[i % 4].run(heads, streams[i % 4])
    dsb.all_gather_rows(torch.zeros((64, 8), device=dev), 64)
    dist.barrier()
    t0 = T()
    n = int(os.environ.get("STEPS", "200"))
    for i in range(n):
        out = engs[i % 4].run(heads, streams[i % 4])
    t_enq = time.perf_counter()
    t1 = T()
    rows, count = out
    with torch.cuda.stream(streams[(n - 1) % 4]):
        g, c = dsb.all_gather_rows(rows, count)
    t2 = T()
    dist.barrier()
    t3 = T()
    print("round %d: %d steps enqueued in %.2f ms, done after %.2f ms; all_gather_rows %.3f ms; barrier %.3f ms"
          % (rnd, n, (t_enq - t0) * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), flush=True)
dist.destroy_process_group()
