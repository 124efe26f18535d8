"""Experiment: replay the three-kernel path from a HIP graph (several batches per graph, each on
its own captured stream) and compare with eager launches."""
import importlib, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
yp = importlib.import_module("2d_object_detection_amd")
syn, util = yp.synthetic, yp.util
dev = torch.device("cuda:0")
B, inp = 64, 416
heads = [h.to(dev) for h in syn.make_heads(B, inp, 5, seed=1)]
for n_eng, reps in ((1, 1), (3, 1), (4, 1), (4, 2), (4, 4)):
    engines = [util.YoloV3Post(B, inp, 5, 0.5, 0.4, device=dev) for _ in range(n_eng)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_eng)]
    for e, s in zip(engines, streams):
        e.run(heads, s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream(device=dev)
    with torch.cuda.graph(g, stream=cap):
        for s in streams:
            s.wait_stream(cap)
        for r in range(reps):
            for e, s in zip(engines, streams):
                e.run(heads, s)
        for s in streams:
            cap.wait_stream(s)
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (n * n_eng * reps)
    ref = engines[0].detect(heads)
    print("graph: %d streams x %d batches each: %.2f us per batch, %.0f images/s (rows %d)"
          % (n_eng, reps, dt * 1e6, B / dt, ref.shape[0]), flush=True)
