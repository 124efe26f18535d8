#!/usr/bin/env python3
"""Diagnostic: where the waves of k_graph_pairs spend their time (in-kernel cycle counters).

Build the diagnostic library first (make -C 2d_object_detection_amd/csrc stamps) and run
    YOLOPOST_LIB=2d_object_detection_amd/libyolopost_stamps.so python tools/graph_stamps.py
B / INP / CONF / CLUSTERED from the environment as for tools/nms_breakdown.py.  The counters are
sums over all waves of the launch; the shipped library contains none of this."""
import ctypes
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("2d_object_detection_amd")
syn, util, _lib = pkg.synthetic, pkg.util, pkg._lib
B = int(os.environ.get("B", 32))
inp = os.environ.get("INP", "1280,1920")
inp = tuple(int(v) for v in inp.split(",")) if "," in inp else int(inp)
conf = float(os.environ.get("CONF", 0.5))
dev = torch.device("cuda:0")
chunks = [syn.make_heads(min(16, B - lo), inp, 5, seed=2000 + lo, clustered=bool(int(os.environ.get("CLUSTERED", "0"))))
          for lo in range(0, B, 16)]
heads = [torch.cat([c[h] for c in chunks], 0).contiguous().to(dev) for h in range(3)]
eng = util.YoloV3Post(B, inp, 5, conf, 0.4, device=dev, dense_nms=True)
nseg = B * 5
dbg = torch.zeros((nseg + 2 + 1280 * 4 // 32 + 1, 32), dtype=torch.int64, device=dev)
L = _lib.load()
L.yp_debug_set_stamps.argtypes = [ctypes.c_void_p]
for _ in range(3):
    eng.run(heads)
torch.cuda.synchronize()
L.yp_debug_set_stamps(dbg.data_ptr())
eng.run(heads)
torch.cuda.synchronize()
L.yp_debug_set_stamps(None)
_lib.profile_start(3)
for _ in range(3):
    eng.run(heads)
ms, calls = _lib.profile_stop_ex()
print("this build: graph_pairs %.1f us per launch" % (1e3 * ms[3]))
c = dbg[nseg].cpu().numpy().astype("float64")
total = c[7]
names = ["batch set-up", "ranges (bin table look-ups, scan)", "locate + issue loads", "wait for the tile's loads",
         "sift + pack", "flush (pair tests, records)", "tail"]
print("B=%d inp=%s conf=%g: batches %d, tiles %d, flush iterations %d; wave-ticks in the batch loops %.3g"
      % (B, inp, conf, c[8], c[9], c[10], total))
for k, n in enumerate(names):
    print("  %-36s %5.1f %%   %8.0f ticks per batch" % (n, 100 * c[k] / total, c[k] / max(1, c[8])))
print("  %-36s %5.1f %%" % ("(unaccounted: batch loop overhead)", 100 * (total - c[:7].sum()) / total))

import numpy as np
w = dbg.reshape(-1)[(nseg + 1) * 32:(nseg + 1) * 32 + 1280 * 4].cpu().numpy().reshape(1280, 4).astype("float64")
w = w[w[:, 0] > 0]
t0 = w[:, 0].min()
st, li, en = (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0, (w[:, 2] - t0) / 100.0
print("workgroups %d: start p50/p90/max %.1f/%.1f/%.1f us; list built after %.1f us (mean); life mean %.1f max %.1f us; "
      "last end %.1f us; segments visited per workgroup mean %.2f max %d"
      % (len(w), np.percentile(st, 50), np.percentile(st, 90), st.max(), (li - st).mean(), (en - st).mean(), (en - st).max(),
         en.max(), w[:, 3].mean(), w[:, 3].max()))
life = en - st
print("life percentiles 10/50/90/99: %s us; end percentiles 10/50/90/99: %s us" % (np.round(np.percentile(life, [10, 50, 90, 99]), 1), np.round(np.percentile(en, [10, 50, 90, 99]), 1)))
for lo in range(0, int(st.max()) + 1, max(1, int(st.max() / 6) + 1)):
    sel = (st >= lo) & (st < lo + max(1, int(st.max() / 6) + 1))
    if sel.any():
        print("  started in [%d, %d) us: %4d workgroups, life mean %.1f max %.1f us, groups mean %.2f" % (lo, lo + max(1, int(st.max() / 6) + 1), sel.sum(), life[sel].mean(), life[sel].max(), w[sel, 3].mean()))

r = dbg[nseg].cpu().numpy().astype("float64")[16:22]
if r[5] > 0:
    print("resolve kernel, per graph segment (%d of them): %.1f rounds, %.1f re-derivation walks of wave 0; wall time (10 ns ticks -> us) "
          "scans %.1f, re-derivations %.1f, output %.1f us" % (r[5], r[0] / r[5], r[1] / r[5], r[2] / r[5] / 100, r[3] / r[5] / 100, r[4] / r[5] / 100))
rr = dbg[nseg].cpu().numpy()[22:32]
print("first graph segment, rounds (columns visited / scan time in us): " + ", ".join("%d / %.1f" % (int(v) & 0xffffffff, (int(v) >> 32) / 100.0) for v in rr if v))
