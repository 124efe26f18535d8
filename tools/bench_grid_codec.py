"""Times the ground-truth grid codec kernels (encode, decode) on one GPU; prints one JSON line per
kernel.  Run it through bench.py, which supplies the CPU baseline leg:

    python bench.py --workload grid_codec

Algorithmic bytes: encode writes the whole grid and mask (slots*(5+2)*4 B per image) and
reads 40 B per object; decode reads the grid once (slots*20 B) and writes 36 B per occupied
slot (index 16 + box 16 + class 4)."""
import argparse
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
yp = importlib.import_module("2d_object_detection_amd")


def run(cpu_baseline=None, argv=None):
    """cpu_baseline(labels_per_image, anchors) -> (encode_seconds, decode_seconds) on the host,
    or None to leave the CPU leg out."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--objects", type=int, default=32)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--cpu-images", type=int, default=64)
    a = ap.parse_args(argv or [])
    rs = np.random.RandomState(0)
    B, n = a.batch, a.objects
    y = np.stack([rs.uniform(0, 1920, B * n), rs.uniform(0, 1280, B * n), rs.uniform(8, 1200, B * n),
                  rs.uniform(8, 900, B * n), rs.randint(0, 5, B * n).astype(np.float64)], 1)
    dev = torch.device("cuda:0")
    rows = torch.from_numpy(y).to(dev)
    offsets = torch.arange(0, (B + 1) * n, n, dtype=torch.int32, device=dev)
    cd = yp.create_dataset
    grid, mask = cd.encode_true_box_grids((rows, offsets))
    idx, boxes, cls, count = cd.decode_true_box_grids_device(grid)
    occupied = int(count.item())
    slots = 13 * 13 * 5

    def timed(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters

    enc_ms = timed(lambda: cd.encode_true_box_grids((rows, offsets)))
    dec_ms = timed(lambda: cd.decode_true_box_grids_device(grid))
    enc_bytes = B * slots * 28 + B * n * 40
    dec_bytes = B * slots * 20 + occupied * 36
    anchors = np.asarray(yp.params.YOLO_ANCHORS)
    m = min(a.cpu_images, B)
    cpu = cpu_baseline([y[b * n:(b + 1) * n] for b in range(m)], anchors) if cpu_baseline else (None, None)
    for name, ms, nbytes, cpu_s in (("grid_encode", enc_ms, enc_bytes, cpu[0]),
                                    ("grid_decode", dec_ms, dec_bytes, cpu[1])):
        print(json.dumps({
            "metric": name + "_images_per_sec", "value": round(B / (ms * 1e-3), 1), "unit": "images/s",
            "ms_per_call": round(ms, 5), "dtype": "f64" if name == "grid_encode" else "f32",
            "config": {"workload": "13x13x5 grid, %d images x %d objects, %d occupied slots"
                       % (B, n, occupied)},
            "roofline": {"bound": "hbm", "achieved": round(nbytes / (ms * 1e-3) / 1e9, 1), "peak": 8000.0,
                         "unit": "GB/s", "frac": round(nbytes / (ms * 1e-3) / 8e12, 4),
                         "algorithmic_bytes": nbytes,
                         "note": "includes the host wrapper (allocation + launch) of each call"},
            "cpu_baseline": None if cpu_s is None else {
                "value": round(m / cpu_s, 1), "unit": "images/s", "cores": 1, "kind": "port",
                "sample": "%d images through the numpy restatement" % m}}))


if __name__ == "__main__":
    run(argv=sys.argv[1:])
