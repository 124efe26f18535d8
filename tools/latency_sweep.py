#!/usr/bin/env python3
"""Latency of one call of the YOLOv3 path for 1 .. 512 images at 416x416 (stream-synchronised,
median of 200 calls), with the per-kernel HIP-event times.  YOLOPOST_LIB selects the build."""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("2d_object_detection_amd")
syn, util, _lib = pkg.synthetic, pkg.util, pkg._lib
dev = torch.device("cuda:0")
inp = int(os.environ.get("INP", 416))
for n_sets, B in ((1, 1), (1, 8), (1, 32), (1, 64), (2, 64), (4, 64), (8, 64)):
    sets = [[h.to(dev) for h in syn.make_heads(B, inp, 5, seed=1 + k)] for k in range(n_sets)]
    eng = util.YoloV3PostMulti(n_sets, B, inp, 5, 0.5, 0.4, device=dev)
    for _ in range(20):
        eng.run(sets)
    torch.cuda.synchronize()
    ts = []
    for _ in range(200):
        t0 = time.perf_counter()
        eng.run(sets)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    _lib.profile_start(50)
    for _ in range(50):
        eng.run(sets)
        torch.cuda.synchronize()
    ms, calls = _lib.profile_stop_ex()
    print("%3d images: call latency median %.1f us (min %.1f); kernels front %.1f nms %.1f gather %.1f us"
          % (n_sets * B, 1e6 * np.median(ts), 1e6 * min(ts), 1e3 * ms[0], 1e3 * ms[1], 1e3 * ms[2]))
