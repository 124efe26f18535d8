"""Experiment: step time of the pipelined loop when only some stages are launched (stage mask in
YpParams.reserved: 1 front, 2 segment NMS, 4 gather)."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
yp = importlib.import_module("2d_object_detection_amd")
syn, util = yp.synthetic, yp.util
dev = torch.device("cuda:0")
B, inp = 64, 416
heads = [h.to(dev) for h in syn.make_heads(B, inp, 5, seed=1)]
n_eng = int(os.environ.get("NF", 3))
engines = [util.YoloV3Post(B, inp, 5, 0.5, 0.4, device=dev) for _ in range(n_eng)]
streams = [torch.cuda.Stream(device=dev) for _ in range(n_eng)]
for e, s in zip(engines, streams):
    e.run(heads, s)
torch.cuda.synchronize()
for mask in (7, 3, 1, 2, 4, 5, 6):
    for e in engines:
        e.params.reserved = mask
    for i in range(30):
        engines[i % n_eng].run(heads, streams[i % n_eng])
    torch.cuda.synchronize()
    n = 600
    t0 = time.perf_counter()
    for i in range(n):
        engines[i % n_eng].run(heads, streams[i % n_eng])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("stages %d: %.2f us per step (host enqueue %.2f us)" % (mask, (t2 - t0) / n * 1e6, (t1 - t0) / n * 1e6), flush=True)
