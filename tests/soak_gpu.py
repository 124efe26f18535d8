#!/usr/bin/env python3
"""Soak run (not collected by pytest): many seeded batches through the HIP path and the CPU
oracle, every output bit compared.  Looks for results that depend on timing (the kernels use LDS
atomics and ballots whose order varies from run to run) rather than on the input.

    python tests/soak_gpu.py [--iters 150] [--seconds 60]
"""
import argparse
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

pkg = importlib.import_module("2d_object_detection_amd")
syn, util = pkg.synthetic, pkg.util


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=150)
    ap.add_argument("--seconds", type=float, default=60.0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    shapes = [(4, 416), (2, 608), (1, (1280, 1920)), (3, (320, 480)), (6, 416)]
    t0, done, rows = time.time(), 0, 0
    for it in range(a.iters):
        if time.time() - t0 > a.seconds:
            break
        B, inp = shapes[it % len(shapes)]
        rs = np.random.RandomState(1000 + it)
        conf = float(rs.choice([0.5, 0.3, 0.7, 0.05]))
        nms = float(rs.choice([0.4, 0.45, 0.3, 0.6]))
        clustered = bool(it % 3 == 1)
        heads = syn.make_heads(B, inp, 5, seed=5000 + it, clustered=clustered)
        want, surv, _ = oracle.n_detect([h.numpy() for h in heads], inp, syn.YOLOV3_ANCHORS, 5, conf, nms)
        eng = util.YoloV3Post(B, inp, 5, conf, nms, device=dev)
        dh = [h.to(dev) for h in heads]
        for rep in range(3):                       # the same input three times: same bits each time
            got = eng.detect(dh).cpu().numpy()
            if got.shape != want.shape or not np.array_equal(got, want):
                print("MISMATCH at iteration %d repetition %d: B=%d inp=%s conf=%g nms=%g clustered=%s "
                      "got %s want %s" % (it, rep, B, inp, conf, nms, clustered, got.shape, want.shape))
                return 1
        done += 1
        rows += want.shape[0]
        if it % 10 == 0:
            print("iteration %d ok (%d rows so far, %.0f s)" % (it, rows, time.time() - t0), flush=True)
    print("soak ok: %d batches, %d detection rows, every bit equal to the oracle" % (done, rows))
    return 0


if __name__ == "__main__":
    sys.exit(main())
