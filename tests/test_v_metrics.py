"""Detections against ground truth (yp_match_detections / train.comp_nms_gt): the comparison
train.py:365-366 describes and the empty comp_nms_gt (train.py:279-280) was meant to hold.
The oracle is the sequential PASCAL-VOC loop of oracle/np_oracle.py; the kernels use the
order-free formulation (every detection finds its candidate, an atomic minimum per ground-truth
box names the first claimant), restated here in numpy to show on the CPU that the two agree."""
import ctypes
import importlib
import os
import sys

import numpy as np
import pytest
import torch

import oracle
from oracle import np_oracle as npo

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
make_golden = importlib.import_module("make_golden")

H = W = 13
A = C = 5
ANCHORS = np.array(((0.57273, 0.677385), (1.87446, 2.06253), (3.33843, 5.47434),
                    (7.88282, 3.52778), (9.77052, 9.16828)))


def golden():
    return np.load(os.path.join(HERE, "golden", "match_cases.npz"))


def claim_formulation(rows, gi, gb, gc, num_classes, thr):
    """The kernels' two passes in numpy: candidates independently, then min list position per
    ground-truth box."""
    thr = np.float32(thr)
    D, G = len(rows), len(gb)
    verdict = np.full(D, -1, np.int32)
    claimed = np.full(G, np.iinfo(np.int32).max, np.int64)
    for d, r in enumerate(rows):
        best_same, best_other, cand = np.float32(0), np.float32(0), -1
        for g in np.flatnonzero(gi[:, 0] == int(r[0])):
            iou = npo.match_iou(r[1:5], gb[g])
            if gc[g] == r[7]:
                if iou > best_same:
                    best_same, cand = iou, g
            elif iou > best_other:
                best_other = iou
        if cand >= 0 and best_same >= thr:
            verdict[d] = cand
            claimed[cand] = min(claimed[cand], d)
        elif best_other >= thr:
            verdict[d] = -3
    counts = np.zeros((5, num_classes), np.int32)
    for d, r in enumerate(rows):
        c = int(r[7])
        if verdict[d] >= 0:
            if claimed[verdict[d]] == d:
                counts[0, c] += 1
                continue
            verdict[d] = -2
        counts[1, c] += 1
        counts[3, c] += verdict[d] == -2
        counts[4, c] += verdict[d] == -3
    for g in np.flatnonzero(claimed == np.iinfo(np.int32).max):
        counts[2, int(gc[g])] += 1
    return verdict, counts


def test_match_oracle_reproduces_golden_and_hand_answers():
    g = golden()
    rows, gi, gb, gc, want_match, want_counts = make_golden.match_hand_case()
    m, iou, counts = npo.match_detections(rows, gi, gb, gc, 5, 0.5)
    assert np.array_equal(m, want_match) and np.array_equal(counts, want_counts)
    assert np.array_equal(m, g["hand_match"]) and np.array_equal(iou, g["hand_iou"])
    assert np.array_equal(rows, g["hand_rows"]) and np.array_equal(gb, g["hand_gt_boxes"])
    for name, thr in (("t05", 0.5), ("t06", 0.6), ("t03", 0.3)):
        m, iou, counts = npo.match_detections(g["rand_rows"], g["rand_gt_index"], g["rand_gt_boxes"],
                                              g["rand_gt_class"], 5, thr)
        assert np.array_equal(m, g["rand_match_" + name])
        assert np.array_equal(counts, g["rand_counts_" + name])
    assert np.array_equal(iou, g["rand_iou"])
    p, r = npo.precision_recall(want_counts)
    assert p[1] == 0.5 and r[1] == 1.0 and r[2] == 0.0 and np.isnan(p[3]) and r[3] == 0.0


@pytest.mark.parametrize("seed", range(5))
def test_first_claim_formulation_equals_the_sequential_loop(seed):
    rows, gi, gb, gc = make_golden.match_random_case(seed=900 + seed, n_images=5)
    for thr in (0.3, 0.5, 0.75):
        m, _, counts = npo.match_detections(rows, gi, gb, gc, 5, thr)
        v, c2 = claim_formulation(rows, gi, gb, gc, 5, thr)
        assert np.array_equal(m, v) and np.array_equal(counts, c2)


def test_match_iou_known_values():
    assert npo.match_iou([0, 0, 10, 10], [5, 5, 10, 10]) == 1.0
    assert npo.match_iou([0, 0, 10, 10], [10, 5, 10, 10]) == np.float32(50.0) / np.float32(150.0)
    assert npo.match_iou([0, 0, 10, 10], [25, 5, 10, 10]) == 0.0
    assert npo.match_iou([0, 0, 0, 0], [0, 0, 0, 0]) == 0.0                      # empty union
    assert npo.match_iou([0, 0, 10, 10], [5, 5, 20, 20]) == 0.25


def test_match_arguments_are_validated_without_a_gpu(pkg):
    L, lib = pkg._lib, pkg._lib.load()
    fake = ctypes.c_void_p(0x1000)
    ok = lambda **kw: lib.yp_match_detections(
        kw.get("rows", fake), fake, kw.get("det_cap", 10), fake, fake, fake, fake, kw.get("gt_cap", 10),
        kw.get("C", 5), kw.get("thr", 0.5), kw.get("match", fake), None, fake, kw.get("ws", fake),
        kw.get("ws_bytes", 1 << 20), None)
    assert ok(rows=None) == -1 and ok(match=None) == -1
    assert ok(det_cap=0) == -1 and ok(gt_cap=0) == -1 and ok(det_cap=(1 << 30) + 1) == -1
    assert ok(C=0) == -1 and ok(C=5000) == -1
    assert ok(thr=0.0) == -1 and ok(thr=1.5) == -1 and ok(thr=float("nan")) == -1
    assert ok(rows=ctypes.c_void_p(0x1004)) == -1                 # rows are read as float4
    assert ok(ws=None) == -2 and ok(ws_bytes=8) == -2 and ok(ws=ctypes.c_void_p(0x1010)) == -2
    assert lib.yp_match_detections_workspace_bytes(10) == 256
    assert lib.yp_match_detections_workspace_bytes(0) == 0
    assert L.YP_MATCH_PLANES == 5


def test_detection_rows_packing(pkg):
    """The padded reference-mode outputs -> the detection list (pure tensor indexing, so it runs
    on CPU tensors too): valid rows first, by image and NMS rank, corners reordered to x1 y1 x2 y2."""
    B, mb = 3, 4
    boxes = torch.arange(B * mb * 4, dtype=torch.float32).view(B, mb, 4)       # [y1, x1, y2, x2]
    scores = torch.rand(B, mb)
    classes = torch.randint(0, 5, (B, mb), dtype=torch.int32)
    counts = torch.tensor([2, 0, 4], dtype=torch.int32)
    rows, n = pkg.train.detection_rows((boxes, scores, classes, None, counts, None))
    assert rows.shape == (B * mb, 8) and int(n) == 6
    want = []
    for b in range(B):
        for k in range(int(counts[b])):
            y1, x1, y2, x2 = boxes[b, k].tolist()
            want.append([b, x1, y1, x2, y2, float(scores[b, k]), float(scores[b, k]), float(classes[b, k])])
    assert np.array_equal(rows[:6].numpy(), np.array(want, np.float32))


def _emulated_library():
    """The library's kernels compiled with g++ against tests/host_emul (every GPU thread a fibre
    on the CPU, barriers and wave collectives as rendezvous): the kernels' own source."""
    import shutil
    import subprocess
    if shutil.which("g++") is None or shutil.which("make") is None:
        pytest.skip("g++ / make are not installed here")
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "host_emul")])
    return ctypes.CDLL(os.path.join(HERE, "_build", "libyolopost_emul.so"))


def _run_emulated(lib, rows, gi, gb, gc, num_classes, thr, det_cap, gt_cap):
    D, G = len(rows), len(gb)

    def pad(a, n, fill, dtype):
        out = np.full((n,) + a.shape[1:], fill, dtype)
        out[:len(a)] = a
        return out
    base = np.zeros(det_cap * 8 + 4, np.float32)                      # a 16-byte aligned view for the float4 loads
    off = (-base.ctypes.data // 4) % 4
    h_rows = base[off:off + det_cap * 8].reshape(det_cap, 8)
    h_rows[:] = 7.0
    h_rows[:D] = rows.reshape(-1, 8)
    gbase = np.zeros(gt_cap * 4 + 4, np.float32)
    goff = (-gbase.ctypes.data // 4) % 4
    h_gb = gbase[goff:goff + gt_cap * 4].reshape(gt_cap, 4)
    h_gb[:] = 50.0
    h_gb[:G] = gb.reshape(-1, 4)
    h_gi, h_gc = pad(gi.reshape(-1, 4), gt_cap, 0, np.int32), pad(gc, gt_cap, 1.0, np.float32)
    nd, ng = np.array([D], np.int32), np.array([G], np.int32)
    match = np.full(det_cap, -77, np.int32)
    iou = np.full(det_cap, -7.0, np.float32)
    counts = np.full((5, num_classes), -5, np.int32)
    lib.yp_match_detections_workspace_bytes.restype = ctypes.c_size_t
    lib.yp_match_detections_workspace_bytes.argtypes = [ctypes.c_int64]
    need = lib.yp_match_detections_workspace_bytes(gt_cap)
    wbase = np.zeros(need + 256, np.uint8)
    woff = (-wbase.ctypes.data) % 256
    vp = ctypes.c_void_p
    lib.yp_match_detections.argtypes = [vp, vp, ctypes.c_int64, vp, vp, vp, vp, ctypes.c_int64, ctypes.c_int32,
                                        ctypes.c_float, vp, vp, vp, vp, ctypes.c_size_t, vp]
    rc = lib.yp_match_detections(h_rows.ctypes.data, nd.ctypes.data, det_cap, h_gb.ctypes.data, h_gc.ctypes.data,
                                 h_gi.ctypes.data, ng.ctypes.data, gt_cap, num_classes, thr, match.ctypes.data,
                                 iou.ctypes.data, counts.ctypes.data, wbase.ctypes.data + woff, need, None)
    assert rc == 0
    assert (match[D:] == -77).all() and (iou[D:] == -7.0).all()
    return match[:D], iou[:D], counts


def test_match_kernels_under_host_emulation():
    """k_match_claim / k_match_count themselves -- the source the GPU compiles -- executed on the
    CPU by the emulation in tests/host_emul, against the oracle: golden cases, scenes with several
    workgroups of detections, empty sides, capacities beyond the counts, and a class count that
    takes the kernels' global-atomics branch (5 x 500 cells exceed the LDS histogram)."""
    lib = _emulated_library()
    g = golden()
    m, iou, counts = _run_emulated(lib, g["hand_rows"], g["hand_gt_index"], g["hand_gt_boxes"], g["hand_gt_class"],
                                   5, 0.5, 6, 4)
    assert np.array_equal(m, g["hand_match"]) and np.array_equal(iou, g["hand_iou"])
    assert np.array_equal(counts, g["hand_counts"])
    for name, thr in (("t05", 0.5), ("t06", 0.6), ("t03", 0.3)):
        m, iou, counts = _run_emulated(lib, g["rand_rows"], g["rand_gt_index"], g["rand_gt_boxes"],
                                       g["rand_gt_class"], 5, thr, 300, 70)
        assert np.array_equal(m, g["rand_match_" + name]) and np.array_equal(counts, g["rand_counts_" + name])
        assert np.array_equal(iou, g["rand_iou"])
    for seed, n_images, C, thr in ((1, 40, 5, 0.5), (4, 60, 500, 0.5), (5, 1, 5, 1.0), (3, 60, 3, 0.25)):
        rows, gi, gb, gc = make_golden.match_random_case(seed=seed, n_images=n_images, num_classes=min(C, 7))
        want = npo.match_detections(rows, gi, gb, gc, C, thr)
        got = _run_emulated(lib, rows, gi, gb, gc, C, thr, len(rows) + 300, len(gb) + 5)
        assert len(rows) > 256 or n_images == 1
        for a, b in zip(got, want):
            assert np.array_equal(a, b), (seed, C)
    rows, gi, gb, gc = make_golden.match_random_case(seed=11, n_images=3)
    m, _, counts = _run_emulated(lib, np.zeros((0, 8), np.float32), gi, gb, gc, 5, 0.5, 1, len(gb))
    assert len(m) == 0 and counts[2].sum() == len(gb) and counts[[0, 1, 3, 4]].sum() == 0
    m, iou, counts = _run_emulated(lib, rows, np.zeros((0, 4), np.int32), np.zeros((0, 4), np.float32),
                                   np.zeros((0,), np.float32), 5, 0.5, len(rows), 1)
    assert (m == -1).all() and (iou == 0).all() and counts[1].sum() == len(rows) and counts[[0, 2]].sum() == 0


# ---------------------------------------------------------------------------------- GPU

def run_abi(pkg, dev, rows, gi, gb, gc, num_classes, thr, det_cap=None, gt_cap=None, want_iou=True):
    """yp_match_detections through ctypes on device copies padded to the given capacities."""
    L, lib = pkg._lib, pkg._lib.load()
    D, G = len(rows), len(gb)
    det_cap = max(D, 1) if det_cap is None else det_cap
    gt_cap = max(G, 1) if gt_cap is None else gt_cap

    def pad(a, n, fill):
        out = np.full((n,) + a.shape[1:], fill, a.dtype)
        out[:len(a)] = a
        return torch.from_numpy(out).to(dev)
    d_rows = pad(rows.reshape(-1, 8), det_cap, 7.0)        # padding rows look like detections
    d_gi, d_gb, d_gc = pad(gi.reshape(-1, 4), gt_cap, 0), pad(gb.reshape(-1, 4), gt_cap, 50.0), pad(gc, gt_cap, 1.0)
    d_nd = torch.tensor([D], dtype=torch.int32, device=dev)
    d_ng = torch.tensor([G], dtype=torch.int32, device=dev)
    match = torch.full((det_cap,), -77, dtype=torch.int32, device=dev)
    iou = torch.full((det_cap,), -7.0, dtype=torch.float32, device=dev)
    counts = torch.full((5, num_classes), -5, dtype=torch.int32, device=dev)
    ws = L.workspace(lib.yp_match_detections_workspace_bytes(gt_cap), dev)
    with L.on_device(dev):
        rc = lib.yp_match_detections(d_rows.data_ptr(), d_nd.data_ptr(), det_cap, d_gb.data_ptr(),
                                     d_gc.data_ptr(), d_gi.data_ptr(), d_ng.data_ptr(), gt_cap, num_classes,
                                     thr, match.data_ptr(), iou.data_ptr() if want_iou else None,
                                     counts.data_ptr(), ws.data_ptr(), ws.numel(), L.stream_ptr(dev))
    assert rc == 0
    torch.cuda.synchronize()
    m, i = match.cpu().numpy(), iou.cpu().numpy()
    assert (m[D:] == -77).all() and (i[D:] == -7.0).all()      # rows beyond the count are untouched
    return m[:D], i[:D], counts.cpu().numpy()


@pytest.mark.gpu
def test_match_gpu_golden(pkg, dev):
    g = golden()
    m, iou, counts = run_abi(pkg, dev, g["hand_rows"], g["hand_gt_index"], g["hand_gt_boxes"],
                             g["hand_gt_class"], 5, 0.5)
    assert np.array_equal(m, g["hand_match"]) and np.array_equal(iou, g["hand_iou"])
    assert np.array_equal(counts, g["hand_counts"])
    for name, thr in (("t05", 0.5), ("t06", 0.6), ("t03", 0.3)):
        m, iou, counts = run_abi(pkg, dev, g["rand_rows"], g["rand_gt_index"], g["rand_gt_boxes"],
                                 g["rand_gt_class"], 5, thr, det_cap=1000, gt_cap=300)
        assert np.array_equal(m, g["rand_match_" + name])
        assert np.array_equal(counts, g["rand_counts_" + name])
        assert np.array_equal(iou, g["rand_iou"])


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_images,num_classes,thr", [
    (1, 40, 5, 0.5), (2, 64, 5, 0.6), (3, 30, 3, 0.25), (4, 20, 500, 0.5), (5, 1, 5, 1.0)])
def test_match_gpu_random_scenes(pkg, dev, seed, n_images, num_classes, thr):
    """More rows than one workgroup holds, class counts beyond the LDS histogram (C = 500)."""
    rows, gi, gb, gc = make_golden.match_random_case(seed=seed, n_images=n_images,
                                                     num_classes=min(num_classes, 7))
    want = npo.match_detections(rows, gi, gb, gc, num_classes, thr)
    got = run_abi(pkg, dev, rows, gi, gb, gc, num_classes, thr, det_cap=len(rows) + 300, gt_cap=len(gb) + 5)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    # without the IoU output
    got = run_abi(pkg, dev, rows, gi, gb, gc, num_classes, thr, want_iou=False)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2])


@pytest.mark.gpu
def test_match_gpu_empty_sides(pkg, dev):
    rows, gi, gb, gc = make_golden.match_random_case(seed=11, n_images=3)
    none = np.zeros((0, 8), np.float32)
    m, _, counts = run_abi(pkg, dev, none, gi, gb, gc, 5, 0.5)
    assert len(m) == 0 and counts[2].sum() == len(gb) and counts[[0, 1, 3, 4]].sum() == 0
    m, iou, counts = run_abi(pkg, dev, rows, np.zeros((0, 4), np.int32), np.zeros((0, 4), np.float32),
                             np.zeros((0,), np.float32), 5, 0.5)
    assert (m == -1).all() and (iou == 0).all() and counts[1].sum() == len(rows) and counts[[0, 2]].sum() == 0


@pytest.mark.gpu
def test_comp_nms_gt_on_the_reference_path(pkg, dev):
    """Raw head -> detect_device -> comp_nms_gt against the oracle's pipeline: r_detect, the
    same row packing, the decoded grid and the sequential matching loop."""
    from test_train_grad import make_case
    train, cd = pkg.train, pkg.create_dataset
    pred, grid = make_case(6, 21)
    pred[:, 4::10] += 6.0
    dets = pkg.DetectNet.detect_device(torch.tensor(pred, device=dev))
    res = train.comp_nms_gt(dets, torch.tensor(grid, device=dev), iou_thr=0.3)
    # oracle side
    boxes, scores, classes, _index = oracle.r_detect(pred, H, W, A, C, ANCHORS, 0.6, 0.5, 20, 1280, 1920)[:4]
    rows = []
    for b in range(6):
        for k in range(len(scores[b])):
            y1, x1, y2, x2 = boxes[b][k]
            rows.append([b, x1, y1, x2, y2, scores[b][k], scores[b][k], classes[b][k]])
    rows = np.array(rows, np.float32).reshape(-1, 8)
    gb, gi = [], []
    for b in range(6):
        bx, lc = npo.decode_true_box_grid(grid[b], ANCHORS, exp=oracle.exp_spec)
        gb.append(bx)
        gi.append(np.concatenate([np.full((len(lc), 1), b, np.int64), lc], 1))
    gb, gi = np.concatenate(gb).astype(np.float32), np.concatenate(gi).astype(np.int32)
    gc = np.array([grid[i[0], i[1], i[2], i[3], 4] for i in gi], np.float32)
    want_m, want_iou, want_c = npo.match_detections(rows, gi, gb, gc, C, 0.3)
    n = int(res.det_count.item())
    assert n == len(rows) > 0 and int(res.gt_count.item()) == len(gb)
    d_rows, _ = train.detection_rows(dets)
    assert np.array_equal(d_rows[:n].cpu().numpy(), rows)
    assert np.array_equal(res.det_match[:n].cpu().numpy(), want_m)
    assert np.array_equal(res.det_iou[:n].cpu().numpy(), want_iou)
    assert np.array_equal(res.class_counts.cpu().numpy(), want_c)
    s = res.summary()
    assert np.array_equal(s["tp"], want_c[0]) and np.array_equal(s["fn"], want_c[2])
    # the grid decoded beforehand, and YOLOv3-style (rows, count) input, give the same result
    res2 = train.comp_nms_gt((d_rows, res.det_count), cd.decode_true_box_grids_device(torch.tensor(grid, device=dev)),
                             iou_thr=0.3)
    assert np.array_equal(res2.class_counts.cpu().numpy(), want_c)


@pytest.mark.gpu
def test_run_validation_accumulates_the_statistics(pkg, dev):
    from test_train_grad import make_case
    pred, grid = make_case(4, 9)
    pred[:, 4::10] += 6.0
    dataset = [(torch.tensor(pred[:2], device=dev), torch.tensor(grid[:2], device=dev)),
               (torch.tensor(pred[2:], device=dev), torch.tensor(grid[2:], device=dev))]
    out = pkg.train.run_validation(dataset, lambda x: x, verbose=False)
    total = np.zeros((5, C), np.int64)
    for k in range(2):
        r = pkg.train.comp_nms_gt(pkg.DetectNet.detect_device(dataset[k][0]), dataset[k][1])
        total += r.class_counts.cpu().numpy()
    assert np.array_equal(out.class_counts, total)
    assert np.array_equal(out.metrics["tp"], total[0]) and np.array_equal(out.metrics["fp"], total[1])
    n_gt = sum(len(npo.true_box_locations(grid[b:b + 1])) for b in range(4))
    assert total[0].sum() + total[2].sum() == n_gt
