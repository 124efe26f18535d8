"""The library's kernels executed on the CPU from their own source (tests/host_emul: every GPU
thread a fibre, wave intrinsics and barriers as rendezvous of fibres) and compared with the
oracle bit for bit -- a check of the kernels' logic (indexing, collectives, barriers, atomics,
float order) that needs no GPU -- and the same under AddressSanitizer + UBSan with every buffer a
heap block of exactly its documented size.  It says nothing about timing or about how the
hardware schedules the code; the GPU parity tests remain the proof that the library works."""
import ctypes
import importlib
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

import oracle
from oracle import np_oracle as npo

HERE = os.path.dirname(os.path.abspath(__file__))
EMUL = os.path.join(HERE, "host_emul")
BUILD = os.path.join(HERE, "_build")
ROOT = os.path.dirname(HERE)


def _make(target):
    if shutil.which("g++") is None or shutil.which("make") is None:
        pytest.skip("g++ / make are not installed here")
    subprocess.check_call(["make", "-s", "-C", EMUL, os.path.join("..", "_build", target)])
    return os.path.join(BUILD, target)


def test_emulation_known_answers():
    """The emulation itself: what ballots, shuffles, readlane / readfirstlane and mbcnt return in
    uniform and in divergent control flow, what barriers order, what the atomics add up to."""
    out = subprocess.run([_make("emul_selftest")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().splitlines()[-1].startswith("ok "), out.stdout[-2000:]


def load_emulated(pkg, target):
    """An emulated build of the library with the argument types of the real one."""
    lib = ctypes.CDLL(_make(target))
    real = pkg._lib.load()
    for name in pkg._lib.EXPORTED_SYMBOLS:
        fn, ref = getattr(lib, name), getattr(real, name)
        fn.restype = ref.restype
        if ref.argtypes is not None:
            fn.argtypes = ref.argtypes
    return lib


@pytest.fixture(scope="module")
def emu(pkg):
    return load_emulated(pkg, "libyolopost_emul.so")


def aligned(nbytes, align=256, fill=0):
    buf = np.full(nbytes + align, fill, np.uint8)
    off = (-buf.ctypes.data) % align
    return buf[off:off + nbytes]


def emul_counters(lib, reset=True):
    a = (ctypes.c_long * 5)()
    lib.emul_stats(a, 1 if reset else 0)
    return dict(zip(("collectives", "divergent", "shuffle_from_inactive", "launches", "workgroups"), a))


def head_descs(pkg, heads, inp_dim, C):
    """YpHead descriptors over host arrays (the emulation reads host memory)."""
    L, syn = pkg._lib, pkg.synthetic
    in_h, in_w = (inp_dim, inp_dim) if np.isscalar(inp_dim) else inp_dim
    out = []
    for h, anc in zip(heads, syn.YOLOV3_ANCHORS):
        gh, gw = h.shape[2], h.shape[3]
        sx, sy = in_w // gw, in_h // gh
        d = L.make_head(None, gh, gw, [tuple(a) for a in oracle.scaled_anchors(anc, sx, sy)], sx, sy)
        d.data = h.ctypes.data
        out.append(d)
    return out


def run_detect(pkg, lib, sets, inp_dim, conf, nms=0.4, C=5, dense=False, frame=None, cap=None, offset=0):
    """yp_detect_yolov3_multi on host arrays; returns (rows per set, counts, stats)."""
    L, util = pkg._lib, pkg.util
    n_sets, B = len(sets), sets[0][0].shape[0]
    flat = [d for heads in sets for d in head_descs(pkg, heads, inp_dim, C)]
    descs = (L.YpHead * len(flat))(*flat)
    p = util._v3_params(B, C, conf, nms, batch_offset=offset, dense=dense, frame=frame)
    need = lib.yp_detect_yolov3_workspace_bytes(descs, 3, n_sets * B, C)
    ws = aligned(need, fill=0xA5)
    rows_per_image = sum(h.shape[1] // (5 + C) * h.shape[2] * h.shape[3] for h in sets[0])
    cap = B * rows_per_image + (B * rows_per_image & 1) if cap is None else cap
    out = aligned(n_sets * cap * 32, 16, fill=0xA5).view(np.float32).reshape(n_sets, cap, 8)
    cnt = np.full(n_sets, -1, np.int32)
    st = np.full((n_sets, 4, B, C), -1, np.int32)
    rc = lib.yp_detect_yolov3_multi(descs, 3, n_sets, ctypes.byref(p), out.ctypes.data, cap, cnt.ctypes.data,
                                    st.ctypes.data, ws.ctypes.data, need, None)
    assert rc == 0, lib.yp_strerror(rc)
    return out, cnt, st


def make_sets(pkg, n_sets, B, inp_dim, seed, clustered=False):
    syn = pkg.synthetic
    return [[np.ascontiguousarray(h.numpy()) for h in syn.make_heads(B, inp_dim, 5, seed=seed + 17 * k, clustered=clustered)]
            for k in range(n_sets)]


@pytest.mark.parametrize("name,n_sets,B,inp_dim,conf,clustered,dense,frame", [
    ("small", 1, 2, 96, 0.3, True, False, None),
    ("cfg2 shape, 3 images", 1, 3, 416, 0.5, False, False, None),
    ("three sets per launch", 3, 2, 96, 0.4, True, False, None),
    ("rectangular input, rescaled to the camera frame", 1, 2, (320, 480), 0.4, False, False, (1280, 1920)),
    ("segments of ~1500: chunked path", 1, 1, 416, 0.01, False, False, None),
    ("segments of ~1500: graph path", 1, 1, 416, 0.01, False, True, None),
    ("clustered segments on the graph path, two sets", 2, 1, (320, 480), 0.01, True, True, None),
    ("graph flag set, no segment large enough", 1, 2, 96, 0.3, True, True, None),
    ("segments of ~7400: LDS-blocked sort in global memory, graph path", 1, 1, (640, 960), 0.001, False, True, None),
])
def test_emulated_detect_matches_oracle(pkg, emu, nms_path, name, n_sets, B, inp_dim, conf, clustered, dense, frame):
    syn = pkg.synthetic
    if nms_path == "image" and ("7400" in name or "two sets" in name or "graph flag set" in name):
        pytest.skip("large segments behind the per-image kernel are covered by the ~1500 cases")
    sets = make_sets(pkg, n_sets, B, inp_dim, 500, clustered)
    emul_counters(emu)
    out, cnt, st = run_detect(pkg, emu, sets, inp_dim, conf, dense=dense, frame=frame, offset=3)
    c = emul_counters(emu)
    # front, [per-image NMS,] per-segment NMS, row gather; with the flag the pair kernel (a second time,
    # for 64-column batches, when an image has room for a segment of 16 384 candidates) and the resolve kernel
    slots_per_image = sum(-(-h.shape[2] * h.shape[3] // 128) * 128 * 3 for h in sets[0])
    assert c["launches"] == 3 + (nms_path == "image") + ((2 + (slots_per_image >= 16384)) if dense else 0), c
    assert c["divergent"] == 0 and c["shuffle_from_inactive"] == 0, c      # every collective was wave-uniform
    for k, heads in enumerate(sets):
        want, surv, _ = oracle.n_detect(heads, inp_dim, syn.YOLOV3_ANCHORS, 5, conf, 0.4)
        if frame is not None:
            want = npo.n_rescale(want, inp_dim, frame)
        want = want.copy()
        want[:, 0] += 3
        assert cnt[k] == len(want), name
        assert np.array_equal(out[k, :len(want)], want), name
        assert np.array_equal(st[k, 0].sum(1), surv) and st[k, 1].sum() == len(want)
        if "path" in name:
            assert st[k, 0].max() > 768          # the case does exercise the large-segment code
        # rows beyond the count were never written
        assert (out[k, len(want):].view(np.uint8) == 0xA5).all()


def test_emulated_graph_path_with_wide_batches(pkg, emu):
    """A segment of more than 16 384 candidates (one class, one 448x672 image, nearly every anchor
    kept by the confidence filter) takes the pair kernel's 64-column batches."""
    syn = pkg.synthetic
    inp_dim, conf, C = (448, 672), 0.001, 1
    heads = [np.ascontiguousarray(h.numpy()) for h in syn.make_heads(1, inp_dim, C, seed=77)]
    emul_counters(emu)
    out, cnt, st = run_detect(pkg, emu, [heads], inp_dim, conf, C=C, dense=True)
    c = emul_counters(emu)
    assert c["launches"] == 6 and c["divergent"] == 0 and c["shuffle_from_inactive"] == 0, c
    want, surv, _ = oracle.n_detect(heads, inp_dim, syn.YOLOV3_ANCHORS, C, conf, 0.4)
    assert st[0, 0].max() >= 16384
    assert cnt[0] == len(want) and np.array_equal(out[0, :len(want)], want)
    assert np.array_equal(st[0, 0].sum(1), surv) and st[0, 1].sum() == len(want)


def test_ab_script_variants_are_valid_builds():
    """Every compile-time variant tools/ab_variants.sh builds for the GPU box parses with its
    defines (host compiler, emulation header, syntax only), and the script itself parses."""
    if shutil.which("g++") is None or shutil.which("bash") is None:
        pytest.skip("g++ / bash are not installed here")
    script = os.path.join(ROOT, "tools", "ab_variants.sh")
    assert subprocess.run(["bash", "-n", script]).returncode == 0
    text = open(script).read()
    assert "tr ',' ' '" in text          # (the separator this test splits the defines at)
    line = [l for l in text.splitlines() if l.startswith("VARIANTS=")]
    assert len(line) == 1
    variants = dict(v.split(":", 1) for v in line[0].split('"')[1].split())
    assert "base" in variants and variants["base"] == ""
    csrc = os.path.join(ROOT, "2d_object_detection_amd", "csrc")
    for name, defs in variants.items():
        defs = [d for d in defs.split(",") if d]
        assert all(d.startswith("-DYP_") for d in defs), (name, defs)
        out = subprocess.run(["g++", "-std=c++20", "-fsyntax-only", "-Wno-attributes", "-I" + EMUL,
                              "-I" + os.path.join(ROOT, "include"), "-I" + csrc] + defs +
                             [os.path.join(EMUL, "emul_yolopost.cpp")], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, (name, out.stderr[-2000:])


def test_emulated_capacity_overflow_and_single_entry(pkg, emu):
    """Fewer output rows than detections: the count is the true total, the slab holds the first
    `cap` rows, nothing is written past it; and yp_detect_yolov3 equals the multi entry with one set."""
    L, util, syn = pkg._lib, pkg.util, pkg.synthetic
    sets = make_sets(pkg, 2, 2, 96, 900, True)
    full, cnt_full, _ = run_detect(pkg, emu, sets, 96, 0.3)
    cap = (int(cnt_full.min()) // 2) & ~1
    out, cnt, _ = run_detect(pkg, emu, sets, 96, 0.3, cap=cap)
    assert np.array_equal(cnt, cnt_full) and cap > 0
    for k in range(2):
        assert np.array_equal(out[k], full[k, :cap])
    descs = (L.YpHead * 3)(*head_descs(pkg, sets[1], 96, 5))
    p = util._v3_params(2, 5, 0.3, 0.4)
    need = emu.yp_detect_yolov3_workspace_bytes(descs, 3, 2, 5)
    ws = aligned(need)
    one = aligned(int(cnt_full[1]) * 32 + 32, 16).view(np.float32).reshape(-1, 8)
    n = np.zeros(1, np.int32)
    rc = emu.yp_detect_yolov3(descs, 3, ctypes.byref(p), one.ctypes.data, len(one), n.ctypes.data, None,
                              ws.ctypes.data, need, None)
    assert rc == 0 and n[0] == cnt_full[1] and np.array_equal(one[:n[0]], full[1, :n[0]])


def test_emulated_write_results_on_decoded_rows(pkg, emu):
    """The rows entry (yp_write_results) on segments above 512 and above 768 candidates, both
    NMS paths, against the oracle on the same decoded rows."""
    from test_gpu_parity import big_segment_rows
    util = pkg.util
    rows, C = big_segment_rows(1000, 7)
    rows = np.ascontiguousarray(rows)
    want, surv, _ = oracle.n_write_results(rows, 0.5, C, 0.05)
    assert surv.max() > 768
    B, N = rows.shape[0], rows.shape[1]
    for dense in (False, True):
        p = util._v3_params(B, C, 0.5, 0.05, dense=dense)
        need = emu.yp_rows_workspace_bytes(B, N, C)
        ws = aligned(need)
        cap = B * N + (B * N & 1)
        out = aligned(cap * 32, 16).view(np.float32).reshape(cap, 8)
        n = np.zeros(1, np.int32)
        rc = emu.yp_write_results(rows.ctypes.data, N, ctypes.byref(p), out.ctypes.data, cap, n.ctypes.data, None,
                                  ws.ctypes.data, need, None)
        assert rc == 0 and n[0] == len(want), dense
        assert np.array_equal(out[:n[0]], want), dense
    c = emul_counters(emu)
    assert c["divergent"] == 0 and c["shuffle_from_inactive"] == 0, c


def test_emulated_write_results_with_equal_scores(pkg, emu):
    """Equal objectness values inside small (image, class) segments: both NMS kernels order them by
    anchor index (the per-image kernel by repeating its 32-bit sort with the slot as tie-break)."""
    util = pkg.util
    rs = np.random.RandomState(21)
    B, N, C = 2, 300, 3
    rows = np.zeros((B, N, 5 + C), np.float32)
    rows[:, :, 0], rows[:, :, 1] = rs.uniform(0, 200, (B, N)), rs.uniform(0, 200, (B, N))
    rows[:, :, 2], rows[:, :, 3] = rs.uniform(8, 60, (B, N)), rs.uniform(8, 60, (B, N))
    rows[:, :, 4] = rs.choice(np.float32([0.6, 0.7, 0.8]), (B, N))
    rows[0, :, 4] = np.float32(0.77)
    rows[np.arange(B)[:, None], np.arange(N)[None, :], 5 + rs.randint(0, C, (B, N))] = 0.9
    rows = np.ascontiguousarray(rows)
    want, surv, _ = oracle.n_write_results(rows, 0.5, C, 0.4)
    assert surv.max() < 512 and want.shape[0] > 50
    p = util._v3_params(B, C, 0.5, 0.4)
    need = emu.yp_rows_workspace_bytes(B, N, C)
    ws = aligned(need, fill=0xA5)
    out = aligned(B * N * 32, 16).view(np.float32).reshape(B * N, 8)
    n = np.zeros(1, np.int32)
    rc = emu.yp_write_results(rows.ctypes.data, N, ctypes.byref(p), out.ctypes.data, B * N, n.ctypes.data, None,
                              ws.ctypes.data, need, None)
    assert rc == 0 and n[0] == len(want)
    assert np.array_equal(out[:n[0]], want)
    c = emul_counters(emu)
    assert c["divergent"] == 0 and c["shuffle_from_inactive"] == 0, c


def test_emulated_write_results_with_a_few_equal_scores(pkg, emu):
    """A handful of candidates of a class share their objectness (groups of 2 to 5), an image with
    35 such pairs, an image with none: ordered by anchor index inside a group on both NMS kernels."""
    from tie_cases import few_ties_rows
    util = pkg.util
    rows = few_ties_rows(N=400)
    B, N, C = rows.shape[0], rows.shape[1], 3
    want, surv, _ = oracle.n_write_results(rows, 0.5, C, 0.4)
    assert want.shape[0] > 50
    p = util._v3_params(B, C, 0.5, 0.4)
    need = emu.yp_rows_workspace_bytes(B, N, C)
    ws = aligned(need, fill=0xA5)
    out = aligned(B * N * 32, 16).view(np.float32).reshape(B * N, 8)
    n = np.zeros(1, np.int32)
    rc = emu.yp_write_results(rows.ctypes.data, N, ctypes.byref(p), out.ctypes.data, B * N, n.ctypes.data, None,
                              ws.ctypes.data, need, None)
    assert rc == 0 and n[0] == len(want)
    assert np.array_equal(out[:n[0]], want)
    c = emul_counters(emu)
    assert c["divergent"] == 0 and c["shuffle_from_inactive"] == 0, c


def test_emulated_reference_mode(pkg, emu):
    """yp_detect_reference (detect_net.py:90-183 in one pass) against the oracle."""
    L, P = pkg._lib, pkg.params
    B, H, W, A, C, mb = 5, 13, 13, 5, 5, 20
    pred = np.ascontiguousarray(pkg.synthetic.make_reference_head(B, seed=77, relu=True, obj_mean=0.5).numpy())
    anc = np.asarray(P.YOLO_ANCHORS, np.float32).reshape(-1, 2)
    head = L.make_head(None, H, W, [tuple(a) for a in anc])
    head.data = pred.ctypes.data
    p = L.YpParams()
    p.mode, p.batch, p.num_classes, p.max_out = L.YP_MODE_REFERENCE, B, C, mb
    p.conf_thr, p.nms_thr, p.im_h, p.im_w = P.object_conf, P.nms_conf, P.im_height, P.im_width
    need = emu.yp_detect_reference_workspace_bytes(B, H, W, A)
    ws = aligned(need)
    boxes = aligned(B * mb * 16, 16).view(np.float32).reshape(B, mb, 4)
    scores = np.zeros((B, mb), np.float32)
    classes, index = np.zeros((B, mb), np.int32), np.zeros((B, mb), np.int32)
    counts, surv = np.zeros(B, np.int32), np.zeros(B, np.int32)
    rc = emu.yp_detect_reference(ctypes.byref(head), ctypes.byref(p), boxes.ctypes.data, scores.ctypes.data,
                                 classes.ctypes.data, index.ctypes.data, counts.ctypes.data, surv.ctypes.data,
                                 ws.ctypes.data, need, None)
    assert rc == 0
    ob, os_, oc, oi, osurv = oracle.r_detect(pred, H, W, A, C, anc, P.object_conf, P.nms_conf, mb, P.im_height,
                                             P.im_width)
    assert np.array_equal(surv, osurv) and counts.sum() > 0
    for b in range(B):
        n = counts[b]
        assert n == len(os_[b])
        assert np.array_equal(boxes[b, :n], ob[b]) and np.array_equal(scores[b, :n], os_[b])
        assert np.array_equal(classes[b, :n], oc[b]) and np.array_equal(index[b, :n], oi[b])
    c = emul_counters(emu)
    assert c["divergent"] == 0 and c["shuffle_from_inactive"] == 0, c


def test_emulated_reference_mode_random_geometry(pkg, emu):
    """yp_detect_reference on random grids (also non-square ones, where the reference's own index
    construction matters), 1..8 anchors, up to 64 classes, thresholds at their extremes and output
    caps from 1 to 5000, against the oracle."""
    L = pkg._lib
    rs = np.random.RandomState(2024)
    for it in range(30):
        H, W = int(rs.randint(1, 30)), int(rs.randint(1, 30))
        A, C = int(rs.randint(1, 9)), int(rs.choice([1, 2, 5, 5, 20, 64]))
        B = int(rs.randint(1, 5))
        mb = int(rs.choice([1, 3, 20, 20, 200, 5000]))
        conf = float(rs.choice([0.0, 0.05, 0.3, 0.6, 0.9]))
        nms = float(rs.choice([0.0, 0.3, 0.5, 0.5, 0.9, 1.0]))
        im_h, im_w = float(rs.choice([1, 416, 1280])), float(rs.choice([1, 416, 1920]))
        pred = rs.normal(rs.choice([-1.0, 0.0, 1.0]), rs.choice([0.5, 1.5, 3.0]), (B, H * W * A * (5 + C)))
        pred = np.ascontiguousarray((np.maximum(pred, 0) if rs.randint(2) else pred).astype(np.float32))
        anc = np.ascontiguousarray(rs.uniform(0.3, 10, (A, 2)).astype(np.float32))
        head = L.make_head(None, H, W, [tuple(a) for a in anc])
        head.data = pred.ctypes.data
        p = L.YpParams()
        p.mode, p.batch, p.num_classes, p.max_out = L.YP_MODE_REFERENCE, B, C, mb
        p.conf_thr, p.nms_thr, p.im_h, p.im_w = conf, nms, im_h, im_w
        need = emu.yp_detect_reference_workspace_bytes(B, H, W, A)
        ws = aligned(need, fill=0xA5)
        boxes = aligned(B * mb * 16, 16).view(np.float32).reshape(B, mb, 4)
        scores = np.zeros((B, mb), np.float32)
        classes, index = np.zeros((B, mb), np.int32), np.zeros((B, mb), np.int32)
        counts, surv = np.zeros(B, np.int32), np.zeros(B, np.int32)
        rc = emu.yp_detect_reference(ctypes.byref(head), ctypes.byref(p), boxes.ctypes.data, scores.ctypes.data,
                                     classes.ctypes.data, index.ctypes.data, counts.ctypes.data, surv.ctypes.data,
                                     ws.ctypes.data, need, None)
        tag = (it, H, W, A, C, B, mb, conf, nms)
        assert rc == 0, tag
        ob, os_, oc, oi, osurv = oracle.r_detect(pred, H, W, A, C, anc, conf, nms, mb, im_h, im_w)
        assert np.array_equal(surv, osurv), tag
        for b in range(B):
            n = counts[b]
            assert n == len(os_[b]), tag
            assert np.array_equal(boxes[b, :n], ob[b]) and np.array_equal(scores[b, :n], os_[b]), tag
            assert np.array_equal(classes[b, :n], oc[b]) and np.array_equal(index[b, :n], oi[b]), tag
    c = emul_counters(emu)
    assert c["divergent"] == 0 and c["shuffle_from_inactive"] == 0, c


def test_emulated_reference_nms_reproduces_tensorflows_published_cases(pkg, emu):
    """The cases of TensorFlow's own unit test of tf.image.non_max_suppression (the routine the
    reference calls, detect_net.py:136) through yp_filter_boxes_reference's kernels."""
    import tf_nms_vectors as tfv
    L = pkg._lib
    for case in tfv.CASES:
        boxes, scores, max_out, thr, want = tfv.arrays(case)
        n = len(scores)
        if n == 0:
            continue                      # (the entry point refuses an image without rows)
        center, wh, obj, cls = tfv.as_decoded_tuple(boxes, scores)
        p = L.YpParams()
        p.mode, p.batch, p.num_classes, p.max_out = L.YP_MODE_REFERENCE, 1, 5, max_out
        p.conf_thr, p.nms_thr, p.im_h, p.im_w = -100.0, thr, 1.0, 1.0
        need = emu.yp_rows_workspace_bytes(1, n, 1)
        ws = aligned(need, fill=0xA5)
        ob = aligned(max_out * 16, 16).view(np.float32).reshape(1, max_out, 4)
        osc = np.zeros((1, max_out), np.float32)
        ocl, oix = np.zeros((1, max_out), np.int32), np.full((1, max_out), -1, np.int32)
        cnt, surv = np.zeros(1, np.int32), np.zeros(1, np.int32)
        rc = emu.yp_filter_boxes_reference(center.ctypes.data, wh.ctypes.data, obj.ctypes.data, cls.ctypes.data, n,
                                           ctypes.byref(p), ob.ctypes.data, osc.ctypes.data, ocl.ctypes.data,
                                           oix.ctypes.data, cnt.ctypes.data, surv.ctypes.data, ws.ctypes.data, need, None)
        assert rc == 0, case[0]
        assert surv[0] == n and cnt[0] == len(want), case[0]
        assert np.array_equal(oix[0, :cnt[0]], want), case[0]
        assert np.array_equal(osc[0, :cnt[0]], scores[want]), case[0]


def write_case(path, pkg, sets, inp_dim, conf, C, flags, cap, frame, expected):
    n_sets, B = len(sets), sets[0][0].shape[0]
    descs = head_descs(pkg, sets[0], inp_dim, C)
    im_h, im_w = (0.0, 0.0) if frame is None else frame
    with open(path, "wb") as f:
        f.write(struct.pack("<8i4f", 0x31455059, n_sets, B, C, len(descs), flags, cap, 0, conf, 0.4, im_h, im_w))
        for d in descs:
            f.write(struct.pack("<3i2f16f", d.grid_h, d.grid_w, d.num_anchors, d.stride_x, d.stride_y, *d.anchors))
        for heads in sets:
            for h in heads:
                f.write(np.ascontiguousarray(h, np.float32).tobytes())
        for want in expected:
            f.write(struct.pack("<i", len(want)))
            f.write(np.ascontiguousarray(want[:cap], np.float32).tobytes())


def replay_all(replays, workers=4):
    """Runs (label, argv, env) replays of recorded calls, a few at a time (they are independent
    single-threaded processes), and returns [(label, CompletedProcess)] in the order given."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=workers) as pool:
        runs = [pool.submit(subprocess.run, argv, capture_output=True, text=True, timeout=900, env=env)
                for _, argv, env in replays]
        return [(label, r.result()) for (label, _, _), r in zip(replays, runs)]


def test_emulated_kernels_under_address_and_ub_sanitizers(pkg, tmp_path):
    """The same kernels, built with -fsanitize=address,undefined, replaying recorded calls whose
    buffers are heap blocks of exactly the documented sizes: small segments, the chunked
    large-segment path, the graph path, several sets with a capacity smaller than the output,
    and the rescale.  The sanitizers must stay silent and the rows must equal the oracle's, also when
    the workgroups of every launch run in another order."""
    syn, L = pkg.synthetic, pkg._lib
    exe = _make("emul_driver_asan")
    cases = [
        ("small", 1, 2, 96, 0.3, True, 0, None, 1.0),
        ("chunked", 1, 1, 416, 0.01, False, 0, None, 1.0),
        ("graph", 1, 1, 416, 0.01, False, L.YP_FLAG_DENSE_NMS, None, 1.0),
        ("overflow", 3, 2, 96, 0.3, True, 0, None, 0.5),
        ("graph overflow rescale", 1, 1, (320, 480), 0.01, True, L.YP_FLAG_DENSE_NMS | L.YP_FLAG_RESCALE,
         (1280.0, 1920.0), 0.3),
    ]
    # (YP_IMG_MIN_IMAGES=1: every launch goes through the per-image NMS kernel first, which finishes the
    # small cases and hands the large ones to the per-segment kernel -- both run under the sanitizers)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=0",
               YP_IMG_MIN_IMAGES="1")
    replays = []
    for k, (name, n_sets, B, inp_dim, conf, clustered, flags, frame, cap_frac) in enumerate(cases):
        sets = make_sets(pkg, n_sets, B, inp_dim, 300, clustered)
        expected = []
        for heads in sets:
            want, _, _ = oracle.n_detect(heads, inp_dim, syn.YOLOV3_ANCHORS, 5, conf, 0.4)
            expected.append(npo.n_rescale(want, inp_dim, frame) if frame is not None else want)
        cap = max(2, int(max(len(w) for w in expected) * cap_frac) + 2 & ~1)
        path = str(tmp_path / ("case%d.bin" % k))
        write_case(path, pkg, sets, inp_dim, conf, 5, flags, cap, frame, expected)
        # workgroups of a launch run in any order on the GPU: the graph path (atomics between
        # workgroups) is replayed with the workgroups last to first and in a scrambled order as well
        for order in (("", "reverse", "shuffle") if name == "graph" else ("",)):
            replays.append((name + " " + order, [exe, path], dict(env, YP_EMUL_ORDER=order)))
    # one segment of 18 000 candidates (a single class): the pair kernel's 64-column batches
    wide = [np.ascontiguousarray(h.numpy()) for h in syn.make_heads(1, (448, 672), 1, seed=77)]
    want, _, _ = oracle.n_detect(wide, (448, 672), syn.YOLOV3_ANCHORS, 1, 0.001, 0.4)
    path = str(tmp_path / "wide.bin")
    write_case(path, pkg, [wide], (448, 672), 0.001, 1, L.YP_FLAG_DENSE_NMS, len(want) + 2 & ~1, None, [want])
    replays.append(("wide batches", [exe, path], dict(env, YP_EMUL_ORDER="reverse")))
    for label, out in replay_all(replays):
        assert out.returncode == 0, (label, out.stdout[-1500:], out.stderr[-3000:])
        assert out.stdout.strip().splitlines()[-1].startswith("ok"), (label, out.stdout[-500:])
        assert "Sanitizer:" not in out.stderr and "runtime error" not in out.stderr, (label, out.stderr[-3000:])


def test_emulated_grid_codec_matches_golden(pkg, emu):
    """yp_encode_true_box_grid / yp_decode_true_box_grid (create_dataset.py:47-115, :294-328) from
    their kernel source on the CPU against the committed vectors."""
    from test_grid_codec import ANCHORS, golden, split_labels, ulp_diff
    g = golden()
    labels = split_labels(g)
    B, H, W, A = len(labels), 13, 13, 5
    rows = np.ascontiguousarray(np.concatenate(labels).astype(np.float64))
    off = np.ascontiguousarray(g["label_offsets"].astype(np.int32))
    anc = np.ascontiguousarray(ANCHORS.astype(np.float32))
    fp = anc.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    grid = np.full((B, H, W, A, 5), 7.0, np.float32)
    mask = np.full((B, H, W, A, 2), 7.0, np.float32)
    rc = emu.yp_encode_true_box_grid(rows.ctypes.data, off.ctypes.data, B, H, W, A, fp, 1280, 1920,
                                     grid.ctypes.data, mask.ctypes.data, None)
    assert rc == 0
    assert np.array_equal(mask, g["mask"]) and np.array_equal(grid[..., [0, 1, 4]], g["grid"][..., [0, 1, 4]])
    assert ulp_diff(grid[..., 2:4], g["grid"][..., 2:4]).max() <= 1     # a binary64 log sits in between
    cap = B * H * W * A
    index = np.zeros((cap, 4), np.int32)
    boxes = aligned(cap * 16, 16).view(np.float32).reshape(cap, 4)
    cls = np.zeros(cap, np.float32)
    n = np.zeros(1, np.int32)
    need = emu.yp_true_box_grid_workspace_bytes(B, H, W, A)
    ws = aligned(need, fill=0xA5)
    gold = np.ascontiguousarray(g["grid"].astype(np.float32))
    rc = emu.yp_decode_true_box_grid(gold.ctypes.data, B, H, W, A, fp, 1280, 1920, index.ctypes.data,
                                     boxes.ctypes.data, cls.ctypes.data, cap, n.ctypes.data, ws.ctypes.data, need, None)
    assert rc == 0 and n[0] == len(g["locations"])
    assert np.array_equal(index[:n[0]], g["locations"]) and np.array_equal(boxes[:n[0]], g["boxes"])
    c = emul_counters(emu)
    assert c["divergent"] == 0 and c["shuffle_from_inactive"] == 0, c


def test_emulated_decode_and_loss_kernels(pkg, emu):
    """predict_transform (detect_net.py:152-183), its backward and the fused loss_custom
    (train.py:123-197) from their kernel source on the CPU: the decode bit-exact against the C
    oracle, loss and gradients against the PyTorch restatement differentiated by autograd."""
    import torch
    from oracle import torch_ref
    from test_train_grad import ANCHORS, A, C, H, W, close, make_case
    pred, grid = make_case(3, 31)
    B = pred.shape[0]
    pred = np.ascontiguousarray(pred.reshape(B, -1))
    anc = np.ascontiguousarray(np.asarray(ANCHORS, np.float32).reshape(A, 2))
    fp = anc.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    center = np.zeros((B, H, W, A, 2), np.float32)
    wh = np.zeros((B, H, W, A, 2), np.float32)
    obj = np.zeros((B, H, W, A, 1), np.float32)
    cls = np.zeros((B, H, W, A, C), np.float32)
    rc = emu.yp_predict_transform_reference(pred.ctypes.data, B, H, W, A, C, fp, center.ctypes.data, wh.ctypes.data,
                                            obj.ctypes.data, cls.ctypes.data, None)
    assert rc == 0
    want = oracle.r_predict_transform(pred, H, W, A, C, anc)
    for got, w in zip((center, wh, obj, cls), want):
        assert np.array_equal(got, w)
    # fused loss + gradient
    tg = np.ascontiguousarray(grid.astype(np.float32))
    loss = np.zeros(4, np.float32)
    gpred = np.zeros_like(pred)
    need = emu.yp_loss_custom_reference_workspace_bytes(B, H, W, A)
    ws = aligned(need, fill=0xA5)
    rc = emu.yp_loss_custom_reference(pred.ctypes.data, tg.ctypes.data, B, H, W, A, C, fp, 5.0, 5.0, 1.0, 0.6,
                                      loss.ctypes.data, gpred.ctypes.data, ws.ctypes.data, need, None)
    assert rc == 0
    p = torch.tensor(pred, requires_grad=True)
    ref = torch_ref.loss_custom(p, torch.tensor(tg), H, W, A, C, ANCHORS, coord_w=5.0, obj_w=5.0, noobj_w=1.0)
    ref[0].backward()
    for k in range(4):      # total, confidence, classification, coordinates
        assert abs(float(loss[k]) - float(ref[k].detach())) <= 2e-5 * abs(float(ref[k].detach())), k
    close(gpred, p.grad.numpy())
    # backward of the decode with random upstream gradients
    rs = np.random.RandomState(3)
    ups = [np.ascontiguousarray(rs.standard_normal(a.shape).astype(np.float32)) for a in (center, wh, obj, cls)]
    gback = np.zeros_like(pred)
    rc = emu.yp_predict_transform_reference_backward(pred.ctypes.data, B, H, W, A, C, fp, *[u.ctypes.data for u in ups],
                                                     gback.ctypes.data, None)
    assert rc == 0
    p2 = torch.tensor(pred, requires_grad=True)
    outs = torch_ref.predict_transform(p2, H, W, A, C, ANCHORS)
    sum((o * torch.tensor(u)).sum() for o, u in zip(outs, ups)).backward()
    close(gback, p2.grad.numpy())
    c = emul_counters(emu)
    assert c["divergent"] == 0 and c["shuffle_from_inactive"] == 0, c


def write_match_case(path, rows, gi, gb, gc, C, thr, det_cap, gt_cap):
    m, iou, counts = npo.match_detections(rows, gi, gb, gc, C, thr)
    with open(path, "wb") as f:
        f.write(struct.pack("<8i4f", 0x32455059, len(rows), len(gb), C, det_cap, gt_cap, 0, 0, thr, 0, 0, 0))
        for a, t in ((rows, np.float32), (gi, np.int32), (gb, np.float32), (gc, np.float32), (m, np.int32),
                     (iou, np.float32), (counts, np.int32)):
            f.write(np.ascontiguousarray(a, t).tobytes())


def test_emulated_matching_kernels_under_sanitizers(pkg, tmp_path):
    """yp_match_detections replayed under AddressSanitizer + UBSan and under ThreadSanitizer: several
    workgroups of detections, capacities beyond the counts, and a class count on the kernels'
    global-atomics branch."""
    import importlib
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    make_golden = importlib.import_module("make_golden")
    exes = [(_make("emul_driver_asan"), {"ASAN_OPTIONS": "detect_leaks=0:detect_stack_use_after_return=0"},
             ("Sanitizer:", "runtime error")),
            (_make("emul_driver_tsan"), {"TSAN_OPTIONS": "halt_on_error=0:report_signal_unsafe=0:exitcode=0"},
             ("ThreadSanitizer",))]
    for seed, n_images, C, thr in ((1, 40, 5, 0.5), (4, 60, 500, 0.5)):
        rows, gi, gb, gc = make_golden.match_random_case(seed=seed, n_images=n_images, num_classes=min(C, 7))
        path = str(tmp_path / "match.bin")
        write_match_case(path, rows, gi, gb, gc, C, thr, len(rows) + 300, len(gb) + 5)
        for exe, extra, forbidden in exes:
            out = subprocess.run([exe, path], capture_output=True, text=True, timeout=600, env=dict(os.environ, **extra))
            assert out.returncode == 0 and out.stdout.strip().splitlines()[-1].startswith("ok"), (out.stdout[-800:],
                                                                                                   out.stderr[-2000:])
            assert not any(w in out.stderr for w in forbidden), out.stderr[-4000:]


def test_emulated_reference_mode_under_sanitizers(pkg, tmp_path):
    """yp_detect_reference -- the reference's own configuration (13 x 13 grid, 5 anchors, conf 0.6,
    NMS 0.5, 20 boxes) on relu'd heads with many survivors -- replayed under AddressSanitizer + UBSan
    and under ThreadSanitizer against the oracle's per-image outputs."""
    P = pkg.params
    B, H, W, A, C, mb = 6, 13, 13, 5, 5, 20
    pred = np.ascontiguousarray(pkg.synthetic.make_reference_head(B, seed=91, relu=True, obj_mean=0.5).numpy())
    anc = np.asarray(P.YOLO_ANCHORS, np.float32).reshape(-1, 2)
    ob, os_, oc, oi, osurv = oracle.r_detect(pred, H, W, A, C, anc, P.object_conf, P.nms_conf, mb, P.im_height,
                                             P.im_width)
    assert sum(len(s) for s in os_) > B            # the case keeps boxes in every image on average
    path = str(tmp_path / "reference.bin")
    with open(path, "wb") as f:
        f.write(struct.pack("<8i4f", 0x33455059, B, H, W, A, C, mb, 0, P.object_conf, P.nms_conf, P.im_height,
                            P.im_width))
        f.write(np.ascontiguousarray(anc, np.float32).tobytes())
        f.write(pred.astype(np.float32).tobytes())
        f.write(np.asarray([len(s) for s in os_], np.int32).tobytes())
        f.write(np.asarray(osurv, np.int32).tobytes())
        for b in range(B):
            for a, t in ((ob[b], np.float32), (os_[b], np.float32), (oc[b], np.int32), (oi[b], np.int32)):
                f.write(np.ascontiguousarray(a, t).tobytes())
    runs = [(_make("emul_driver_asan"), {"ASAN_OPTIONS": "detect_leaks=0:detect_stack_use_after_return=0"},
             ("Sanitizer:", "runtime error")),
            (_make("emul_driver_tsan"), {"TSAN_OPTIONS": "halt_on_error=0:report_signal_unsafe=0:exitcode=0"},
             ("ThreadSanitizer",))]
    for exe, extra, forbidden in runs:
        out = subprocess.run([exe, path], capture_output=True, text=True, timeout=600, env=dict(os.environ, **extra))
        assert out.returncode == 0 and out.stdout.strip().splitlines()[-1].startswith("ok"), (out.stdout[-800:],
                                                                                               out.stderr[-2000:])
        assert not any(w in out.stderr for w in forbidden), out.stderr[-4000:]


def test_emulated_kernels_under_thread_sanitizer(pkg, tmp_path):
    """The kernels once more, built with -fsanitize=thread: every fibre is a thread to the sanitizer
    and every barrier / wave collective a release-acquire among its participants, so two threads of
    a workgroup that meet at an LDS or global address (one of them writing) with no barrier or
    collective in between are reported as a data race -- the kind of bug that the GPU shows only
    now and then.  Small segments, the chunked path and the graph path must replay without a report;
    and with the workgroup barriers hidden from the sanitizer (self-test) reports must appear, which
    shows that the detection is alive."""
    syn, L = pkg.synthetic, pkg._lib
    exe = _make("emul_driver_tsan")
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0:report_signal_unsafe=0:exitcode=0", YP_IMG_MIN_IMAGES="1")
    cases = [("small", 2, 2, 96, 0.3, True, 0), ("chunked", 1, 1, 416, 0.01, False, 0),
             ("graph", 1, 1, 416, 0.01, False, L.YP_FLAG_DENSE_NMS)]
    replays = []
    for name, n_sets, B, inp_dim, conf, clustered, flags in cases:
        sets = make_sets(pkg, n_sets, B, inp_dim, 300, clustered)
        expected = [oracle.n_detect(heads, inp_dim, syn.YOLOV3_ANCHORS, 5, conf, 0.4)[0] for heads in sets]
        cap = max(len(w) for w in expected) + 2 & ~1
        path = str(tmp_path / ("%s.bin" % name))
        write_case(path, pkg, sets, inp_dim, conf, 5, flags, cap, None, expected)
        replays.append((name, [exe, path], env))
    replays.append(("self-test", [exe, str(tmp_path / "small.bin")], dict(env, YP_EMUL_TSAN_SELFTEST="1")))
    results = replay_all(replays)
    for name, out in results[:-1]:
        assert out.returncode == 0 and out.stdout.strip().splitlines()[-1].startswith("ok"), (name, out.stdout[-800:])
        assert "ThreadSanitizer" not in out.stderr, (name, out.stderr[-4000:])
    assert results[-1][1].stderr.count("WARNING: ThreadSanitizer: data race") >= 5, results[-1][1].stderr[-2000:]
