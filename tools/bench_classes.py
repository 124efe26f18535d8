#!/usr/bin/env python3
"""Diagnostic: the fused path at other class counts (the generic front kernel instead of the
5-class single-wave one).  python tools/bench_classes.py [C ...]"""
import importlib
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # 4 streams on queues of their own (see bench.py)

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("2d_object_detection_amd")
syn, util, _lib = pkg.synthetic, pkg.util, pkg._lib
dev = torch.device("cuda:0")
for C in [int(a) for a in sys.argv[1:]] or [5, 20, 80]:
    B = 64
    heads = [h.to(dev) for h in syn.make_heads(B, 416, C, seed=1)]
    engs = [util.YoloV3Post(B, 416, C, 0.5, 0.4, device=dev) for _ in range(4)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
    for i in range(40):
        engs[i % 4].run(heads, streams[i % 4])
    torch.cuda.synchronize()
    n = 400
    t0 = time.perf_counter()
    for i in range(n):
        engs[i % 4].run(heads, streams[i % 4])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    for i in range(20):
        engs[0].run(heads)
    _lib.profile_start(100)
    for i in range(100):
        engs[0].run(heads)
    f, m, g, _ = _lib.profile_stop()
    nbytes = 4.0 * (5 + C) * B * 10647
    print("C=%d: %.0f images/s, %.1f us per batch; alone: front %.1f us (%.2f of the HBM roofline on %.0f MB), "
          "NMS %.1f us, gather %.1f us" % (C, B / dt, dt * 1e6, f * 1e3, nbytes / (f * 1e-3) / 8e12,
                                          nbytes / 1e6, m * 1e3, g * 1e3))
