#!/usr/bin/env python3
"""Diagnostic: host-side cost of enqueueing one batch (Python wrapper, ctypes call, HIP launches).
The queue is kept short (a synchronise every 64 calls) so that the numbers are enqueue costs, not
back-pressure from a full queue."""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("2d_object_detection_amd")
syn, util = pkg.synthetic, pkg.util
dev = torch.device("cuda:0")
heads = [h.to(dev) for h in syn.make_heads(64, 416, 5, seed=1)]
eng = util.YoloV3Post(64, 416, 5, 0.5, 0.4, device=dev)
eng.run(heads)
torch.cuda.synchronize()


def timed(fn, n=2048):
    t = 0.0
    for blk in range(n // 64):
        t0 = time.perf_counter()
        for _ in range(64):
            fn()
        t += time.perf_counter() - t0
        torch.cuda.synchronize()
    return t / n * 1e6


for mask, name in ((0, "all three kernels"), (1, "front kernel only"), (2, "NMS kernel only"),
                   (4, "row gather only")):
    eng.params.reserved = mask
    print("%-20s run(): %.2f us per call" % (name, timed(lambda: eng.run(heads))))
eng.params.reserved = 0
key = tuple(h.data_ptr() for h in heads)
descs = eng._desc_cache[key]
st = torch.cuda.current_stream(dev).cuda_stream
print("%-20s raw ctypes call: %.2f us" % ("all three kernels", timed(lambda: eng._detect(
    descs, len(descs), eng._params_ref, eng._out_ptr, eng.cap, eng._count_ptr, eng._stats_ptr,
    eng._ws_ptr, eng._ws_bytes, st))))
