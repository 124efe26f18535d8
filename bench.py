#!/usr/bin/env python3
"""Benchmark of the YOLO post-processing hot path (decode -> confidence filter -> NMS).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg1|cfg3|cfg4|cfg5]

One "step" = one pass of the fused path (three HIP kernels through the C ABI of
libyolopost.so) over one batch of synthetic raw detection heads that are already resident
in HBM.  At N=1 the default workload is BASELINE.json's configs[1]: batch 64 at 416x416,
three YOLOv3 heads x 3 anchors, 5 classes, conf 0.5 / NMS 0.4.  With N>1 (launched by
torch.distributed.run, one rank per GPU) every rank processes its own batch of the same size
(weak scaling).  Images are independent, so a step needs no collective; the ranks exchange
their variable-length detection lists with one RCCL all-gather at the end (--exchange final,
the default; one untimed exchange beforehand warms the collective up) or, opt-in and
experimental, after every step (--exchange every-step, overlapped with the next steps; at
4 M images/s per GPU that is ~100 GB/s of rows per rank to replicate everywhere, more than
xGMI carries, so that mode is communication-bound by construction).

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      the front kernel (decode + filter + compaction), algorithmic bytes
                40*B*N read + 28*sum(K) + 4*tiles written, divided by its HIP-event time
  cpu_baseline  the CPU oracle (oracle/ypo.c, a single-threaded port of the same path)
                timed on this host on the same workload
"""
import argparse
import importlib
import json
import os
import sys
import time

# HIP gives a process 4 hardware queues by default, one of which serves the null stream, so at
# most 3 streams run concurrently; the path is fastest with 4 batches in flight on 4 queues of
# their own.  Must be set before the HIP runtime initialises.  (YP_HW_QUEUES overrides.)
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("YP_HW_QUEUES", "8")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("2d_object_detection_amd")
syn, util, dsb, _lib = pkg.synthetic, pkg.util, pkg.distributed, pkg._lib

CONFIGS = {
    # name: (batch per GPU, input size, conf, nms, seed, description)
    "cfg1": (1, 416, 0.5, 0.4, 0, "1 image 416x416 (10,647 anchors), conf 0.5 / NMS 0.4"),
    "cfg2": (64, 416, 0.5, 0.4, 1, "batch 64 at 416x416 (10,647 anchors/image), conf 0.5 / NMS 0.4"),
    "cfg3": (32, (1280, 1920), 0.5, 0.4, 2,
             "batch 32 at 1920x1280 (151,200 anchors/image), conf 0.5 / NMS 0.4"),
    "cfg3x2": (64, (1280, 1920), 0.5, 0.4, 2,
               "batch 64 at 1920x1280 (387 MB of heads, larger than the 256 MiB Infinity Cache)"),
    "cfg4": (64, 608, 0.5, 0.4, 3, "batch 64 per GPU at 608x608 (22,743 anchors/image), conf 0.5 / NMS 0.4"),
    "cfg5": (32, (1280, 1920), 0.001, 0.4, 4,
             "batch 32 per GPU at 1920x1280, conf 0.001 / NMS 0.4 (near-all survive)"),
}
HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak
NUM_CLASSES = 5


def make_device_heads(batch, inp_dim, seed, dev, clustered):
    """Seeded CPU generation in chunks (bounded host memory), uploaded once."""
    chunks, step = [], 16
    for lo in range(0, batch, step):
        n = min(step, batch - lo)
        chunks.append(syn.make_heads(n, inp_dim, NUM_CLASSES, seed=seed * 1000 + lo, clustered=clustered))
    cpu = [torch.cat([c[h] for c in chunks], 0).contiguous() for h in range(3)]
    return cpu, [h.to(dev) for h in cpu]


def front_kernel_roofline(config_name, dev, launches=30):
    """The front kernel timed alone (start/stop HIP events attached to each launch) on another
    configuration's shape: same kernel, same accounting as the headline's roofline object."""
    batch, inp_dim, conf, nms, seed, desc = CONFIGS[config_name]
    _, hs = make_device_heads(batch, inp_dim, seed, dev, False)
    eng = util.YoloV3Post(batch, inp_dim, NUM_CLASSES, conf, nms, device=dev)
    for _ in range(3):
        eng.run(hs)
    _lib.profile_start(launches)
    for _ in range(launches):
        eng.run(hs)
    front_ms, _, _, calls = _lib.profile_stop()
    torch.cuda.synchronize()
    surv, _, _ = eng.segment_stats()
    n_anchors = batch * syn.anchors_per_image(inp_dim if isinstance(inp_dim, int) else tuple(inp_dim))
    tiles = batch * sum(-(-(gh * gw) // 128) for gh, gw in eng.grids)
    alg = 40.0 * n_anchors + 28.0 * int(surv.sum()) + 4.0 * tiles
    achieved = alg / (front_ms * 1e-3) / 1e9
    return {"workload": "%s: %s" % (config_name, desc), "bound": "hbm", "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
            "algorithmic_bytes": int(alg), "kernel_ms": round(front_ms, 5), "launches": calls,
            "regime": "one batch at a time on one stream (as frac_alone)"}


def cpu_baseline(cpu_heads, inp_dim, conf, nms, budget_s=10.0):
    """The CPU oracle on the same workload, single thread; repeats until ~budget_s."""
    import oracle
    hs = [h.numpy() for h in cpu_heads]
    B = hs[0].shape[0]
    sample = min(B, 64)
    hs = [h[:sample] for h in hs]
    reps, t0 = 0, time.perf_counter()
    while True:
        oracle.n_detect(hs, inp_dim, syn.YOLOV3_ANCHORS, NUM_CLASSES, conf, nms)
        reps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or reps >= 200:
            break
    return {"value": round(sample * reps / el, 2), "unit": "images/s", "cores": 1, "kind": "port",
            "sample": "%d images of the same workload x %d repetitions (%.1f s), oracle/ypo.c "
                      "single thread" % (sample, reps, el)}


def side_workload(args):
    """The grid codec / training kernels (tools/bench_*.py) with their CPU baseline legs: the
    restatements under oracle/ timed on the host, handed to the tools as plain callables."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    if args.workload == "reference":
        import bench_reference_mode as tool

        def cpu(pred):
            import oracle
            P = pkg.params
            anchors = np.asarray(P.YOLO_ANCHORS, np.float32).reshape(-1, 2)
            t0 = time.perf_counter()
            oracle.r_detect(pred, P.grid_height, P.grid_width, P.num_anchors, P.num_classes, anchors,
                            P.object_conf, P.nms_conf, P.max_boxes, P.im_height, P.im_width)
            return time.perf_counter() - t0

        return tool.run(cpu_baseline=None if args.no_cpu_baseline else cpu)
    if args.workload == "grid_codec":
        import bench_grid_codec as tool

        def cpu(labels, anchors):
            from oracle import np_oracle as npo
            t0 = time.perf_counter()
            grids = [npo.create_true_box_grid(y, anchors)[0] for y in labels]
            t1 = time.perf_counter()
            for g in grids:
                npo.decode_true_box_grid(g, anchors)
            return t1 - t0, time.perf_counter() - t1
    else:
        import bench_train_kernels as tool

        def cpu(pred, grid, anchors):
            from oracle import torch_ref
            t0 = time.perf_counter()
            p = pred.clone().requires_grad_(True)
            torch_ref.loss_custom(p, grid, 13, 13, 5, 5, anchors)[0].backward()
            return time.perf_counter() - t0
    tool.run(None if args.no_cpu_baseline else cpu)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--clustered", action="store_true",
                    help="overlapping boxes around object centres instead of i.i.d. boxes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=10.0,
                    help="CPU work spent on the cpu_baseline leg (N = 1 only)")
    ap.add_argument("--exchange", default="final", choices=["final", "every-step"],
                    help="N > 1: when the ranks exchange detection rows (see timed_run)")
    ap.add_argument("--workload", default="detect",
                    choices=["detect", "reference", "grid_codec", "train_kernels"],
                    help="detect = the headline path (default); reference = the reference's own 13x13 configuration; "
                         "the others time the label-side and "
                         "training-side kernels and print one JSON line per kernel")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="independent batches in flight (engines x HIP streams); 1 = latency mode; "
                         "0 (default) = 3 or 4, whichever a short untimed probe finds faster here")
    ap.add_argument("--no-large-input", action="store_true",
                    help="skip the extra front-kernel measurement at the cfg3 shape")
    ap.add_argument("--also-other-exchange", action="store_true",
                    help="N > 1: also time the exchange schedule that was not selected, on a short loop")
    args = ap.parse_args()
    if args.workload != "detect":
        return side_workload(args)

    if os.environ.get("YP_BENCH_WATCHDOG"):
        # diagnostic: after that many seconds dump every thread's Python stack and exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["YP_BENCH_WATCHDOG"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal on a single GPU: take the multi-rank code path (process group, barriers, final
    # exchange) with a world of one rank
    multi = world > 1 or os.environ.get("YP_BENCH_FORCE_DIST") == "1"
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch N>1 with torch.distributed.run"
                         % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (there is no CPU path to benchmark)")
    local_rank %= torch.cuda.device_count()      # (rehearsals may put several ranks on one GPU)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")   # (torch.distributed.run sets both)
        backend = os.environ.get("YP_BENCH_BACKEND", "nccl")   # nccl = RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    batch, inp_dim, conf, nms, seed, desc = CONFIGS[args.config]
    cpu_heads, heads = make_device_heads(batch, inp_dim, seed + 100 * rank, dev, args.clustered)
    rows_per_image = syn.anchors_per_image(inp_dim if isinstance(inp_dim, int) else tuple(inp_dim))

    # n_eng engines (own workspace and output buffers), each on its own HIP stream: consecutive
    # batches are independent, so the decode of batch i+1 overlaps the NMS of batch i.
    streams = [torch.cuda.Stream(device=dev) for _ in range(max(4, args.in_flight))]

    def probe(n, steps=150):
        """Local steps per second with n batches in flight (untimed calibration)."""
        engs = [util.YoloV3Post(batch, inp_dim, NUM_CLASSES, conf, nms, device=dev) for _ in range(n)]
        for i in range(30):
            engs[i % n].run(heads, streams[i % n])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            engs[i % n].run(heads, streams[i % n])
        torch.cuda.synchronize()
        return steps / (time.perf_counter() - t0)

    if args.in_flight > 0:
        n_eng = args.in_flight
    else:
        rates = {n: probe(n) for n in (3, 4)}
        n_eng = max(rates, key=rates.get)

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_run(exchange, steps, warmup):
        """K steps of the path on this rank's batches.  exchange (N > 1 only):
        "final"      detections stay sharded by image while the steps run; the ranks exchange the
                     variable-length detection list of the last batch with one all-gather at the
                     end of the timed region
        "every-step" one fixed-capacity all-gather of every rank's rows after every step, enqueued
                     behind the kernels and overlapped with the following steps"""
        gathers, engines = [], []
        per_step = multi and exchange == "every-step"

        def build(gather_cap):
            gathers.clear()
            engines.clear()
            for _ in range(n_eng):
                if per_step:   # the engine writes straight into the payload of its gather buffer
                    g = dsb.PaddedRowGather(gather_cap, 8, dev)     # rows of 32 B
                    gathers.append(g)
                    engines.append(util.YoloV3Post(batch, inp_dim, NUM_CLASSES, conf, nms, device=dev,
                                                   batch_offset=rank * batch, out=g.rows))
                else:
                    engines.append(util.YoloV3Post(batch, inp_dim, NUM_CLASSES, conf, nms, device=dev,
                                                   batch_offset=rank * batch))

        gather_cap = batch * 2048
        build(gather_cap)
        if per_step:
            # size the fixed-capacity exchange from one real batch: largest per-rank detection
            # count over all ranks + 25 %, so the collective moves little more than the payload
            _, cnt = engines[0].run(heads)
            torch.cuda.synchronize()
            cmax = cnt.to(torch.int64).clone()
            dist.all_reduce(cmax, op=dist.ReduceOp.MAX)
            gather_cap = max(1024, -(-int(cmax.item()) * 5 // 4 // 1024) * 1024)
            build(gather_cap)
        pending = [None] * n_eng
        last = [None]
        torch.cuda.synchronize()

        def step(i):
            k = i % n_eng
            if not per_step:
                last[0] = engines[k].run(heads, streams[k])
                return
            with torch.cuda.stream(streams[k]):
                if pending[k] is not None:
                    pending[k].wait()                # stream-level wait: buffer k is free again
                    pending[k] = None
                _, count = engines[k].run(heads, streams[k])
                pending[k] = gathers[k].gather(count, async_op=True)

        def drain():
            for k in range(n_eng):
                if pending[k] is not None:
                    with torch.cuda.stream(streams[k]):
                        pending[k].wait()
                    pending[k] = None

        for i in range(warmup):
            step(i)
        drain()
        if multi and not per_step:
            # the exchange is warmed up like the kernels, with a real batch's detection list: the
            # first RCCL collective of a given size class costs 30-130 ms (measured,
            # tools/exp_exchange_cost.py), later ones 0.2-0.3 ms
            if last[0] is None:
                last[0] = engines[0].run(heads, streams[0])
            for _ in range(2):
                with torch.cuda.stream(streams[0]):
                    streams[0].wait_stream(torch.cuda.current_stream(dev))
                    torch.cuda.synchronize()
                    dsb.all_gather_rows(last[0][0], last[0][1])
        fence()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        t_enqueued = time.perf_counter() - t0
        drain()
        info = {}
        if multi and not per_step:
            rows, count = last[0]
            with torch.cuda.stream(streams[(steps - 1) % n_eng]):   # behind the last batch's kernels
                gathered, counts = dsb.all_gather_rows(rows, count)
            info = {"gather_rows_last_step": counts, "gathered_rows": int(gathered.shape[0])}
        fence()
        elapsed = time.perf_counter() - t0
        if multi:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        if per_step:
            gathered, counts = gathers[(steps - 1) % n_eng].unpack()
            assert len(counts) == world
            info = {"gather_rows_capacity_per_rank": gather_cap, "gather_rows_last_step": counts,
                    "gather_overflow": bool(max(counts) > gather_cap)}
            dist.barrier()
        return elapsed, t_enqueued, engines, info

    elapsed, t_enqueued, engines, xinfo = timed_run(args.exchange, args.steps, args.warmup)
    other = None
    if multi and args.also_other_exchange:
        # the other exchange mode on a shorter loop, reported next to the headline
        other_mode = "every-step" if args.exchange == "final" else "final"
        o_steps = max(10, min(args.steps, 60))
        o_el, _, _, o_info = timed_run(other_mode, o_steps, min(args.warmup, 5))
        other = {"exchange": other_mode, "steps": o_steps,
                 "images_per_s": round(world * batch * o_steps / o_el, 1)}
        other.update({k: v for k, v in o_info.items() if k != "gather_rows_last_step"})

    # per-kernel execution times: each launch carries its own start/stop HIP events on the launch
    # stream (separate pass so the event bookkeeping does not perturb the throughput above)
    nprof = max(10, min(args.steps, 200))
    # (a) in the regime of the timed region: the same engines and streams, batches in flight
    for i in range(4 * len(engines)):
        engines[i % len(engines)].run(heads, streams[i % len(engines)])
    _lib.profile_start(nprof)
    for i in range(nprof):
        engines[i % len(engines)].run(heads, streams[i % len(engines)])
    front_ms, nms_ms, gather_ms, calls = _lib.profile_stop()
    torch.cuda.synchronize()
    # (b) one batch at a time on one stream, after 50 untimed calls to settle into that regime
    for i in range(50):
        engines[0].run(heads)
    torch.cuda.synchronize()
    _lib.profile_start(nprof)
    for i in range(nprof):
        engines[0].run(heads)
    front_alone_ms, nms_alone_ms, gather_alone_ms, _ = _lib.profile_stop()
    torch.cuda.synchronize()

    # the front kernel alone, as a train of launches between two events on one stream: its
    # launch-to-launch period (kernel execution + the dispatch gap between dependent launches)
    eng0 = engines[0]
    eng0.params.reserved = 1
    ntrain = 50
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        eng0.run(heads)
    ev0.record()
    for _ in range(ntrain):
        eng0.run(heads)
    ev1.record()
    torch.cuda.synchronize()
    front_train_ms = ev0.elapsed_time(ev1) / ntrain
    eng0.params.reserved = 0
    eng0.run(heads)
    torch.cuda.synchronize()

    surv, kept, pairs = engines[0].segment_stats()
    total_surv, total_kept = int(surv.sum()), int(kept.sum())

    if rank == 0:
        n_anchors = batch * rows_per_image
        tiles = batch * sum(-(-(gh * gw) // 128) for gh, gw in engines[0].grids)
        # algorithmic bytes of the front kernel: every anchor's 10 f32 logits read once,
        # 28 B written per survivor (box 16, objectness 4, class score 4, class id 4) and one
        # int32 survivor count per 128-cell tile
        alg_bytes = 40.0 * n_anchors + 28.0 * total_surv + 4.0 * tiles
        achieved = alg_bytes / (front_ms * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "r01_front_kernel_traffic.json")
        if os.path.exists(tfile) and not args.clustered:
            # HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of this
            # same command, collected offline and committed; see the file for the correction
            traffic = json.load(open(tfile)).get(args.config, {}).get("traffic_bytes")
        out = {
            "metric": "decode+NMS images/sec",
            "value": round(world * batch * args.steps / elapsed, 1),
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (seeded N(0,1) offsets, N(0,0.5^2) sizes, N(-3,2^2) objectness "
                    "logits%s; inputs resident in HBM)" % (", clustered" if args.clustered else ""),
            "config": {"workload": "%s: %s; YOLOv3 heads stride 32/16/8 x 3 anchors, %d classes, "
                                   "NCHW f32" % (args.config, desc, NUM_CLASSES),
                       "batch_per_gpu": batch, "global_batch": batch * world,
                       "anchors_per_image": rows_per_image,
                       "parallelism": "dp%d (images sharded over the ranks, no collective inside a step)"
                                      % world if world > 1 else "dp1",
                       "survivors_per_image": round(total_surv / batch, 1),
                       "kept_per_image": round(total_kept / batch, 1),
                       "batches_in_flight": n_eng,
                       "hw_queues_requested": int(os.environ["GPU_MAX_HW_QUEUES"]),
                       "host_enqueue_us_per_step": round(t_enqueued / args.steps * 1e6, 2)},
            "roofline": {"bound": "hbm", "kernel": "k_front_heads_v3 (decode+filter+compaction)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "algorithmic_bytes": int(alg_bytes),
                         "kernel_ms": round(front_ms, 5),
                         "kernel_ms_method": "mean over %d launches of the start/stop HIP events "
                                             "attached to each launch on its stream, %d batches in "
                                             "flight as in the timed region" % (calls, n_eng),
                         "kernel_ms_alone": round(front_alone_ms, 5),
                         "frac_alone": round(alg_bytes / (front_alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                         "launch_train_period_ms": round(front_train_ms, 5),
                         "whole_path_frac": round(alg_bytes / (elapsed / args.steps) / 1e9
                                                  / HBM_PEAK_GBPS, 4)},
            "kernels_ms": {"front": round(front_ms, 5), "segment_nms": round(nms_ms, 5),
                           "gather": round(gather_ms, 5), "calls": calls,
                           "alone": {"front": round(front_alone_ms, 5), "segment_nms": round(nms_alone_ms, 5),
                                     "gather": round(gather_alone_ms, 5)},
                           "nms_iou_pairs": int(pairs.astype(np.int64).sum()),
                           "nms_iou_pairs_per_s": round(float(pairs.astype(np.int64).sum())
                                                        / (nms_ms * 1e-3), 1)},
        }
        # NMS against the vector units' peak: one IoU test is ~22 VALU instructions per pair
        # (4 min/max, 2 sub+add, 2 clamps, product, union, 2 band products, 3-4 compares and
        # selects), the chip issues 256 CUs x 64 lanes per cycle at 2.4 GHz
        ops_per_pair, lane_ops_peak = 22.0, 256 * 64 * 2.4e9
        pairs_per_s = float(pairs.astype(np.int64).sum()) / (nms_ms * 1e-3)
        out["nms_roofline"] = {"bound": "valu", "achieved": round(pairs_per_s / 1e9, 2),
                               "peak": round(lane_ops_peak / ops_per_pair / 1e9, 1), "unit": "G IoU pairs/s",
                               "frac": round(pairs_per_s * ops_per_pair / lane_ops_peak, 4),
                               "ops_per_pair": ops_per_pair,
                               "note": "pairs the kernel evaluated / its HIP-event time; the kernel also "
                                       "gathers, sorts and sweeps, so this is a lower bound on the tile rate"}
        if multi:
            out["config"]["exchange"] = (
                "final: detections stay sharded by image during the steps; one variable-length RCCL "
                "all-gather of the last batch's detection rows closes the timed region"
                if args.exchange == "final" else
                "every-step: one fixed-capacity RCCL all-gather of all ranks' detection rows per step, "
                "overlapped with the following steps")
            out["config"].update(xinfo)
            if other is not None:
                out["config"]["other_exchange_mode"] = other
        if world == 1 and args.config == "cfg2" and not args.clustered and not args.no_large_input:
            # the same kernel on the path's largest single-GPU shape, where the launch is long
            # enough for the fixed dispatch / drain cost of a 9 us kernel not to dominate
            del engines
            out["roofline_large_input"] = front_kernel_roofline("cfg3", dev)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cpu_heads, inp_dim, conf, nms, args.cpu_baseline_seconds)
        print(json.dumps(out), flush=True)
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
