#!/usr/bin/env python3
"""Diagnostic: where the per-image NMS workgroups spend their time (in-kernel stamps).

Build the diagnostic library first (make -C 2d_object_detection_amd/csrc stamps) and run
    YOLOPOST_LIB=2d_object_detection_amd/libyolopost_stamps.so python tools/img_stamps.py
Stamps are 100 MHz wall-clock ticks taken by thread 0 of every workgroup; the shipped
library contains none of this."""
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
SETS = int(os.environ.get("SETS", 8))
inp = int(os.environ.get("INP", "416"))
CONF = float(os.environ.get("CONF", 0.5))
dev = torch.device("cuda:0")
sets = [[h.to(dev) for h in syn.make_heads(B, inp, 5, seed=1 + k)] for k in range(SETS)]
eng = util.YoloV3PostMulti(SETS, B, inp, 5, CONF, 0.4, device=dev)
nseg = SETS * B * 5
dbg = torch.zeros((nseg, 32), dtype=torch.int64, device=dev)
L = _lib.load()
L.yp_debug_set_stamps.argtypes = [ctypes.c_void_p]
for _ in range(5):
    eng.run(sets)
L.yp_debug_set_stamps(dbg.data_ptr())
eng.run(sets)
torch.cuda.synchronize()
L.yp_debug_set_stamps(None)
n_img = SETS * B
raw = dbg.cpu().numpy()[:n_img]
ok = (raw[:, :7] > 0).all(1)
print("workgroups with all seven stamps: %d of %d" % (ok.sum(), n_img))
t = raw[ok].astype(np.float64) * 10.0 / 1000.0   # ticks of 10 ns -> us
names = ["gather", "strips", "sort", "boxes", "pairs", "sweep"]
t0 = t[:, 0].min()
print("start times, percentiles 50/90/99: %s us" % np.round(np.percentile(t[:, 0] - t[:, 0].min(), [50, 90, 99]), 2))
print("images %d; last start %.2f us, last end %.2f us (relative to the first start)"
      % (n_img, (t[:, 0] - t0).max(), (t[:, 6] - t0).max()))
for k, n in enumerate(names):
    d = t[:, k + 1] - t[:, k]
    print("%-8s mean %.2f us  max %.2f us" % (n, d.mean(), d.max()))
print("workgroup total mean %.2f us max %.2f us" % ((t[:, 6] - t[:, 0]).mean(), (t[:, 6] - t[:, 0]).max()))
tot = t[:, 6] - t[:, 0]
print("workgroup total percentiles 50/90/99/max: %s us" % np.round(np.percentile(tot, [50, 90, 99, 100]), 2))
for k, n in enumerate(names):
    d = t[:, k + 1] - t[:, k]
    print("  %-8s percentiles 50/90/99/max: %s us" % (n, np.round(np.percentile(d, [50, 90, 99, 100]), 2)))
print("the five slowest workgroups (phases in us):")
for i in np.argsort(-tot)[:5]:
    print("  total %.2f: %s" % (tot[i], ", ".join("%s %.2f" % (n, t[i, k + 1] - t[i, k]) for k, n in enumerate(names))))
