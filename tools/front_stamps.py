#!/usr/bin/env python3
"""Diagnostic: timeline of the heads front kernel's workgroups (in-kernel stamps, diagnostic
build only).  YOLOPOST_LIB=.../libyolopost_stamps.so python tools/front_stamps.py"""
import ctypes
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("2d_object_detection_amd")
syn, util, _lib = pkg.synthetic, pkg.util, pkg._lib
B = int(os.environ.get("B", 64))
dev = torch.device("cuda:0")
heads = [h.to(dev) for h in syn.make_heads(B, 416, 5, seed=1)]
eng = util.YoloV3Post(B, 416, 5, 0.5, 0.4, device=dev)
nblk = B * 30
dbg = torch.zeros((nblk, 4), dtype=torch.int64, device=dev)
L = _lib.load()
L.yp_debug_set_front_stamps.argtypes = [ctypes.c_void_p]
for _ in range(5):
    eng.run(heads)
L.yp_debug_set_front_stamps(dbg.data_ptr())
eng.run(heads)
torch.cuda.synchronize()
L.yp_debug_set_front_stamps(None)
t = dbg.cpu().numpy().astype(np.float64) * 0.01     # 10 ns ticks -> us
t -= t[:, 0].min()
print("workgroups %d" % nblk)
for k, n in enumerate(["start", "loaded", "ranked", "end"]):
    print("%-7s min %.2f  median %.2f  p95 %.2f  max %.2f us" % (
        n, t[:, k].min(), np.median(t[:, k]), np.percentile(t[:, k], 95), t[:, k].max()))
d = t[:, 1:] - t[:, :-1]
print("per workgroup: load wait %.2f, filter+rank+stage %.2f, dense decode+store %.2f us (medians)"
      % tuple(np.median(d, 0)))
