"""Times the reference's own configuration of the path (detect_net.py:90-183 with the params.py
defaults: one 13x13 head, 5 anchors, 5 classes, softmax, score >= 0.6, class-agnostic NMS at 0.5,
at most 20 boxes) on one GPU; prints one JSON line per batch size.  Run it through bench.py,
which supplies the CPU baseline leg:

    python bench.py --workload reference

Three figures per batch size: the drop-in two-call surface the reference's callers use
(predict_transform, then filter_boxes, results as numpy arrays: includes the host copies and
synchronisation the return type forces), the fused call that leaves its results on the device
(DetectNet.detect_device, no synchronisation inside the loop), and the two kernels' own times.
Algorithmic bytes of the front kernel: 40 B read per anchor, 28 B written per survivor, 4 B per
256-anchor tile."""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
yp = importlib.import_module("2d_object_detection_amd")


def run(cpu_baseline=None, argv=None):
    """cpu_baseline(pred numpy (B, 8450)) -> seconds for one pass over it on the host, or None."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="16,4096")
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--cpu-images", type=int, default=256)
    a = ap.parse_args(argv or [])
    dev = torch.device("cuda:0")
    DN, P, L = yp.DetectNet, yp.params, yp._lib
    n_anchor = P.grid_height * P.grid_width * P.num_anchors
    for B in [int(x) for x in a.batches.split(",")]:
        cpu_pred = yp.synthetic.make_reference_head(B, seed=5, relu=True, obj_mean=0.5)
        pred = cpu_pred.to(dev)

        def two_call():
            return DN.filter_boxes(DN.predict_transform(pred))

        for _ in range(3):
            two_call()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_two = max(3, min(a.iters, 20))
        for _ in range(n_two):
            two_call()
        two_s = (time.perf_counter() - t0) / n_two

        for _ in range(5):
            outs = DN.detect_device(pred)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            outs = DN.detect_device(pred)
        torch.cuda.synchronize()
        fused_s = (time.perf_counter() - t0) / a.iters
        L.profile_start(a.iters)
        for _ in range(a.iters):
            outs = DN.detect_device(pred)
        front_ms, nms_ms, _, calls = L.profile_stop()
        torch.cuda.synchronize()
        surv = int(outs[5].sum().item())
        kept = int(outs[4].sum().item())
        tiles = B * -(-n_anchor // 256)
        alg = 40.0 * B * n_anchor + 28.0 * surv + 4.0 * tiles
        cpu = None
        if cpu_baseline:
            m = min(a.cpu_images, B)
            cpu_s = cpu_baseline(cpu_pred[:m].numpy())
            cpu = {"value": round(m / cpu_s, 1), "unit": "images/s", "cores": 1, "kind": "port",
                   "sample": "%d images through oracle/ypo.c, single thread" % m}
        print(json.dumps({
            "metric": "reference-mode decode+NMS images/sec", "value": round(B / fused_s, 1),
            "unit": "images/s", "ms_per_call": round(fused_s * 1e3, 5), "dtype": "f32",
            "config": {"workload": "params.py defaults: batch %d, 13x13 grid x 5 anchors x (5+5), "
                                   "conf %.2f, NMS %.2f, max %d boxes; relu'd N(0,1) head"
                                   % (B, P.object_conf, P.nms_conf, P.max_boxes),
                       "survivors_per_image": round(surv / B, 2), "kept_per_image": round(kept / B, 2)},
            "two_call_drop_in": {"images_per_s": round(B / two_s, 1), "ms_per_call": round(two_s * 1e3, 4),
                                 "note": "predict_transform + filter_boxes returning numpy arrays "
                                         "(host copies and synchronisation included)"},
            "kernels_ms": {"front": round(front_ms, 5), "segment_nms": round(nms_ms, 5), "calls": calls},
            "roofline": {"bound": "hbm", "kernel": "k_front_grid_ref (decode+softmax+filter+compaction)",
                         "achieved": round(alg / (front_ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(alg / (front_ms * 1e-3) / 8e12, 4), "traffic": None,
                         "algorithmic_bytes": int(alg)},
            "cpu_baseline": cpu}), flush=True)


if __name__ == "__main__":
    run(argv=sys.argv[1:])
