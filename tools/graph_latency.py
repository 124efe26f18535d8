#!/usr/bin/env python3
"""Latency of one call of the YOLOv3 path, eager against the same call captured into a HIP graph
and replayed (torch.cuda.CUDAGraph: the library's launches go to the capturing stream, it allocates
nothing and never synchronises, so a call can be captured as it is).  1, 8 and 64 images of
416x416; median and minimum of 300 stream-synchronised calls; the replayed result is compared
with the eager one."""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("2d_object_detection_amd")
syn, util, _lib = pkg.synthetic, pkg.util, pkg._lib
dev = torch.device("cuda:0")
for B in (1, 8, 64):
    heads = [h.to(dev) for h in syn.make_heads(B, 416, 5, seed=1)]
    eng = util.YoloV3Post(B, 416, 5, 0.5, 0.4, device=dev)
    for _ in range(20):
        eng.run(heads)
    torch.cuda.synchronize()
    def lat(fn, n=300):
        ts = []
        for _ in range(n):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        return 1e6 * np.median(ts), 1e6 * min(ts)
    eager = lat(lambda: eng.run(heads))
    ref_rows, ref_cnt = eng.run(heads)
    ref = ref_rows[:int(ref_cnt)].clone()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        eng.run(heads)
    torch.cuda.current_stream().wait_stream(s)
    try:
        with torch.cuda.graph(g):
            out = eng.run(heads)
        g.replay(); torch.cuda.synchronize()
        rows, cnt = out
        same = int(cnt) == ref.shape[0] and torch.equal(rows[:int(cnt)], ref)
        rep = lat(g.replay)
        print("B=%d: eager %.1f us (min %.1f), graph replay %.1f us (min %.1f), same result %s" % (B, *eager, *rep, same))
    except Exception as e:
        print("B=%d: eager %.1f us; capture failed: %r" % (B, eager[0], e))
