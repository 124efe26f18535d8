#!/usr/bin/env python3
"""Diagnostic: HIP-event time of every kernel of the path (front, segment NMS, row gather, graph
pair tests, graph resolve), one batch at a time.  B / INP / CONF / CLUSTERED / DENSE (0 or 1: force the
large-segment path off or on) from the environment."""
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
dense = None if "DENSE" not in os.environ else bool(int(os.environ["DENSE"]))      # default: the engine decides
eng = util.YoloV3Post(B, inp, 5, conf, 0.4, device=dev, dense_nms=dense)
for _ in range(3):
    eng.run(heads)
torch.cuda.synchronize()
n = int(os.environ.get("N", 10))
_lib.profile_start(n)
for _ in range(n):
    eng.run(heads)
ms, calls = _lib.profile_stop_ex()
surv, kept, pairs = eng.segment_stats()
print("B=%d inp=%s conf=%g dense=%s: front %.1f  nms %.1f  gather %.1f  graph_pairs %.1f  graph_resolve %.1f us; "
      "survivors/segment mean %.0f max %d, kept %d, pairs %.3g"
      % (B, inp, conf, eng.dense_nms, *(1e3 * v for v in ms), surv.mean(), surv.max(), kept.sum(), float(pairs.astype('int64').sum())))
