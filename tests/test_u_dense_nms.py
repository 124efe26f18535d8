"""YP_FLAG_DENSE_NMS: large (image, class) segments finished by the graph kernels (a spatial
index, pair tests spread over many workgroups, fixed-point resolution of the keep set) instead of
by the segment's one workgroup.  The flag changes how the keep set is computed, never what it is:
every output bit must equal the oracle's and the other path's."""
import numpy as np
import pytest
import torch

import oracle
from test_gpu_parity import big_segment_rows

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,seed,nms,degenerate,offset", [
    (1500, 1, 0.4, False, 0.0),
    (3000, 2, 0.45, False, 0.0),
    (4700, 8, 0.4, False, 0.0),
    (9000, 9, 0.3, False, 0.0),       # several tiles of candidate rows per batch of columns
    (1400, 3, 0.4, True, 0.0),        # boxes without a positive size: the segment declines the path
    (1300, 4, 0.0, False, 0.0),       # thresholds outside [0.05, 1]: likewise
    (1300, 5, 1.0, False, 0.0),       # threshold 1: only identical boxes suppress
    (1600, 6, 0.4, False, 3.0e7),     # coordinates where a float's spacing exceeds 2 px
    (1000, 7, 0.05, False, 0.0),      # just above the path's minimum size, lowest threshold it takes
])
def test_dense_nms_rows_match_oracle_and_default_path(pkg, dev, n, seed, nms, degenerate, offset):
    util = pkg.util
    rows, C = big_segment_rows(n, seed, degenerate=degenerate, offset=offset)
    want, surv, _ = oracle.n_write_results(rows, 0.5, C, nms)
    assert surv.max() > 768
    d_rows = torch.from_numpy(rows).to(dev)
    for dense in (True, False):
        got = util.write_results(d_rows, 0.5, C, nms, dense_nms=dense)
        got = np.zeros((0, 8), np.float32) if isinstance(got, int) else got.cpu().numpy()
        assert got.shape == want.shape, (dense, got.shape, want.shape)
        assert np.array_equal(got, want), dense
    # statistics: survivors and kept rows per segment agree between the two paths
    _, _, st_d = util.write_results_device(d_rows, 0.5, C, nms, want_stats=True, dense_nms=True)
    _, _, st_o = util.write_results_device(d_rows, 0.5, C, nms, want_stats=True, dense_nms=False)
    assert np.array_equal(st_d[:2].cpu().numpy(), st_o[:2].cpu().numpy())


@pytest.mark.parametrize("inp_dim,B,conf,clustered,seed", [
    (416, 2, 0.001, False, 4),              # near-all survive: ~2000 candidates per segment
    ((320, 480), 2, 0.01, True, 12),
    ((1280, 1920), 1, 0.5, True, 13),       # clustered scene at full size: dense overlaps
    ((1280, 1920), 1, 0.001, False, 14),    # ~30k candidates per segment
])
def test_dense_nms_from_heads_matches_oracle(pkg, dev, inp_dim, B, conf, clustered, seed):
    syn, util = pkg.synthetic, pkg.util
    heads = syn.make_heads(B, inp_dim, 5, seed=seed, clustered=clustered)
    want, surv, _ = oracle.n_detect([h.numpy() for h in heads], inp_dim, syn.YOLOV3_ANCHORS, 5, conf, 0.4)
    dheads = [h.to(dev) for h in heads]
    for dense in (True, False):
        eng = util.YoloV3Post(B, inp_dim, 5, conf, 0.4, device=dev, dense_nms=dense)
        got = eng.detect(dheads).cpu().numpy()
        assert got.shape == want.shape, (dense, got.shape, want.shape)
        assert np.array_equal(got, want), dense
        s_surv, s_kept, _ = eng.segment_stats()
        assert np.array_equal(s_surv.sum(1), surv) and int(s_kept.sum()) == want.shape[0]
        assert np.array_equal(eng.detect(dheads).cpu().numpy(), got)      # again, same buffers


def test_dense_flag_is_chosen_from_the_configuration(pkg, dev):
    util = pkg.util
    assert util.YoloV3Post(1, 416, 5, 0.001, 0.4, device=dev).dense_nms is True
    assert util.YoloV3Post(1, 416, 5, 0.5, 0.4, device=dev).dense_nms is False
    assert util.YoloV3Post(1, 96, 5, 0.001, 0.4, device=dev).dense_nms is False       # few anchors
    assert util.YoloV3Post(1, 416, 5, 0.5, 0.4, device=dev, dense_nms=True).dense_nms is True


def test_cfg5_full_batch_properties(pkg, dev):
    """The whole per-GPU batch of BASELINE.json's cfg5 (32 images at 1920x1280, conf 0.001: about
    29,500 candidates per (image, class)) -- too large for the oracle in a test, so checked through
    properties that do not depend on the size: both NMS paths give the same bytes; images are
    independent (the first image alone gives the same rows, and those equal the oracle's); rows
    are ordered image -> class -> objectness; counts add up; and inside a segment no two kept
    boxes exceed the threshold (what greedy NMS guarantees; the quotient is formed with the same
    binary32 operations in the same order as the kernel's exact test, so equality is the limit)."""
    import os
    syn, util = pkg.synthetic, pkg.util
    B = int(os.environ.get("YP_TEST_CFG5_BATCH", "32"))
    inp_dim, conf, thr = (1280, 1920), 0.001, 0.4
    chunks = [syn.make_heads(min(16, B - lo), inp_dim, 5, seed=4000 + lo) for lo in range(0, B, 16)]
    heads = [torch.cat([c[h] for c in chunks], 0).contiguous() for h in range(3)]
    dheads = [h.to(dev) for h in heads]
    eng = util.YoloV3Post(B, inp_dim, 5, conf, thr, device=dev)
    assert eng.dense_nms is True
    got = eng.detect(dheads).clone()
    surv, kept, _ = eng.segment_stats()
    assert surv.shape == (B, 5) and surv.min() > 768
    assert int(kept.sum()) == got.shape[0] and (kept <= surv).all() and (kept > 0).all()
    # the one-workgroup-per-segment path: same bytes
    other = util.YoloV3Post(B, inp_dim, 5, conf, thr, device=dev, dense_nms=False).detect(dheads)
    assert torch.equal(got, other)
    del other
    # order: image, then class, then objectness descending; every row passed the filter
    img, cls, obj = got[:, 0], got[:, 7], got[:, 5]
    key = img.to(torch.int64) * 8 + cls.to(torch.int64)
    assert bool((key[1:] >= key[:-1]).all())
    same = key[1:] == key[:-1]
    assert bool((obj[1:][same] <= obj[:-1][same]).all())
    assert bool((obj > conf).all())
    counts = torch.bincount(key, minlength=B * 8).view(B, 8)[:, :5].cpu().numpy()
    assert np.array_equal(counts, kept)
    # images are independent: image 0 alone gives the same rows, and they are the oracle's
    first = util.YoloV3Post(1, inp_dim, 5, conf, thr, device=dev).detect([h[:1].contiguous() for h in dheads])
    n0 = int(kept[0].sum())
    assert torch.equal(first, got[:n0])
    want, _, _ = oracle.n_detect([h[:1].numpy() for h in heads], inp_dim, syn.YOLOV3_ANCHORS, 5, conf, thr)
    assert np.array_equal(first.cpu().numpy(), want)
    # no two kept boxes of a segment exceed the threshold (pixel-inclusive IoU, as the path defines it)
    for b, c in ((0, 0), (B - 1, 4), (B // 2, 2)):
        seg = got[(img == b) & (cls == c)][:, 1:5]
        n = seg.shape[0]
        area = (seg[:, 2] - seg[:, 0] + 1) * (seg[:, 3] - seg[:, 1] + 1)
        worst = 0.0
        for lo in range(0, n, 2048):
            a = seg[lo:lo + 2048]
            iw = (torch.minimum(a[:, None, 2], seg[None, :, 2]) - torch.maximum(a[:, None, 0], seg[None, :, 0]) + 1).clamp(min=0)
            ih = (torch.minimum(a[:, None, 3], seg[None, :, 3]) - torch.maximum(a[:, None, 1], seg[None, :, 1]) + 1).clamp(min=0)
            inter = iw * ih
            iou = inter / (area[lo:lo + 2048, None] + area[None, :] - inter)
            iou[torch.arange(a.shape[0]), torch.arange(lo, lo + a.shape[0])] = 0.0      # a box with itself
            worst = max(worst, float(iou.max()))
        assert worst <= thr, (b, c, worst)


def _write_results_abi(pkg, dev, rows, C, conf, nms, dense, max_out=0):
    """yp_write_results through the C ABI with an output cap per segment (YpParams.max_out)."""
    import ctypes
    L, util = pkg._lib, pkg.util
    lib = L.load()
    B, N, _ = rows.shape
    d_rows = torch.from_numpy(rows).to(dev)
    p = util._v3_params(B, C, conf, nms, dense=dense)
    p.max_out = max_out
    ws = L.workspace(lib.yp_rows_workspace_bytes(B, N, C), dev)
    out = torch.full((B * N, 8), -1.0, dtype=torch.float32, device=dev)
    count = torch.zeros((1,), dtype=torch.int32, device=dev)
    stats = torch.zeros((4, B, C), dtype=torch.int32, device=dev)
    rc = lib.yp_write_results(d_rows.data_ptr(), N, ctypes.byref(p), out.data_ptr(), B * N, count.data_ptr(),
                              stats.data_ptr(), ws.data_ptr(), ws.numel(), L.stream_ptr(dev))
    assert rc == 0, lib.yp_strerror(rc)
    torch.cuda.synchronize()
    return out[:int(count.item())].cpu().numpy(), stats.cpu().numpy()


def test_dense_nms_on_a_score_monotone_chain(pkg, dev):
    """A run of boxes, each overlapping only its neighbours above the threshold, with scores
    falling along the run: box 0 is kept, suppresses box 1, box 2 is kept because its only
    suppressor is gone, and so on -- the keep set of every box hangs on the one before it, the
    longest dependency chain a segment of that size can have for the resolve kernel's rounds."""
    import time
    n, C = 1500, 2
    rows = np.zeros((1, n, 5 + C), np.float32)
    rows[0, :, 0] = 50.0 + 30.0 * np.arange(n)          # centres 30 px apart, boxes 100 px wide:
    rows[0, :, 1] = 40.0                                # IoU 0.54 with the neighbour, 0.25 with the next
    rows[0, :, 2], rows[0, :, 3] = 100.0, 60.0
    rows[0, :, 4] = np.linspace(0.999, 0.501, n).astype(np.float32)
    rows[0, :, 5] = 0.9
    want, surv, _ = oracle.n_write_results(rows, 0.5, C, 0.4)
    assert surv.max() == n and want.shape[0] == n // 2
    for dense in (True, False):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got, _ = _write_results_abi(pkg, dev, rows, C, 0.5, 0.4, dense)
        el = time.perf_counter() - t0
        assert np.array_equal(got, want), dense
        assert el < 2.0, (dense, el)                    # (hundreds of rounds, each a few microseconds)


def test_dense_nms_with_an_output_cap_per_segment(pkg, dev):
    """YpParams.max_out > 0 on the graph path: every (image, class) segment ends after its first
    max_out kept rows, exactly as on the default path."""
    rows, C = big_segment_rows(2500, 11)
    full, _ = _write_results_abi(pkg, dev, rows, C, 0.5, 0.4, dense=False)
    for cap in (1, 7, 300):
        got_d, st_d = _write_results_abi(pkg, dev, rows, C, 0.5, 0.4, dense=True, max_out=cap)
        got_o, st_o = _write_results_abi(pkg, dev, rows, C, 0.5, 0.4, dense=False, max_out=cap)
        assert np.array_equal(got_d, got_o), cap
        # ... which is the head of every segment of the uncapped result
        want = np.concatenate([full[full[:, 7] == c][:cap] for c in range(C)])
        assert np.array_equal(got_d, want), cap
        assert np.array_equal(st_d[1], st_o[1]) and st_d[1].max() <= cap
