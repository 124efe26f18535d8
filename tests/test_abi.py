"""CPU tests of the C ABI: the library loads, exports every symbol include/yolopost.h
declares, and rejects bad arguments before touching the GPU."""
import ctypes
import os
import re

import pytest
import torch


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg._lib.load()
    header = open(os.path.join(os.path.dirname(__file__), "..", "include", "yolopost.h")).read()
    declared = set(re.findall(r"\b(yp_[a-z0-9_]+)\s*\(", header))
    assert declared == set(pkg._lib.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.yp_version() >= 200
    assert lib.yp_strerror(0) == b"ok"
    assert b"workspace" in lib.yp_strerror(-2)


def test_struct_layout_matches_header(pkg):
    assert ctypes.sizeof(pkg._lib.YpHead) == 8 + 4 * 4 + 2 * 4 + 16 * 4
    assert ctypes.sizeof(pkg._lib.YpParams) == 40


def test_workspace_sizes(pkg):
    L, lib = pkg._lib, pkg._lib.load()
    heads = (L.YpHead * 3)(*[L.make_head(None, g, g, [(1, 1)] * 3, 416 // g, 416 // g)
                             for g in (13, 26, 52)])
    small = lib.yp_detect_yolov3_workspace_bytes(heads, 3, 1, 5)
    big = lib.yp_detect_yolov3_workspace_bytes(heads, 3, 64, 5)
    # 30 tiles of 384 slots per image, 69 bytes of scratch per slot
    assert small >= 30 * 384 * 69 and small % 256 == 0
    assert 64 * small >= big > 32 * small
    assert lib.yp_detect_yolov3_workspace_bytes(heads, 0, 1, 5) == 0
    assert lib.yp_detect_yolov3_workspace_bytes(heads, 3, 0, 5) == 0
    assert lib.yp_rows_workspace_bytes(16, 845, 1) > 0
    assert lib.yp_rows_workspace_bytes(16, 0, 1) == 0
    # the reference-mode entries leave out the arrays of the graph path (YOLOv3 mode only)
    assert 0 < lib.yp_detect_reference_workspace_bytes(16, 13, 13, 5) < lib.yp_rows_workspace_bytes(16, 845, 1)
    assert lib.yp_detect_reference_workspace_bytes(16, 13, 13, 5) == lib.yp_detect_reference_workspace_bytes(16, 845, 1, 1)


def test_bad_arguments_are_rejected_without_a_gpu(pkg):
    L, lib = pkg._lib, pkg._lib.load()
    p = L.YpParams()
    p.mode, p.batch, p.num_classes, p.conf_thr, p.nms_thr = L.YP_MODE_YOLOV3, 1, 5, 0.5, 0.4
    fake = ctypes.c_void_p(0x1000)
    heads = (L.YpHead * 1)(L.make_head(None, 13, 13, [(1, 1)] * 3, 32, 32))
    # null data pointer
    assert lib.yp_detect_yolov3(heads, 1, ctypes.byref(p), fake, 10, fake, None, fake, 1 << 30, None) == -1
    heads[0].data = 0x1000
    # workspace too small / misaligned
    assert lib.yp_detect_yolov3(heads, 1, ctypes.byref(p), fake, 10, fake, None, fake, 16, None) == -2
    assert lib.yp_detect_yolov3(heads, 1, ctypes.byref(p), fake, 10, fake, None,
                                ctypes.c_void_p(0x1010), 1 << 30, None) == -2
    # wrong mode for the entry point, too many anchors, bad batch
    p.mode = L.YP_MODE_REFERENCE
    assert lib.yp_detect_yolov3(heads, 1, ctypes.byref(p), fake, 10, fake, None, fake, 1 << 30, None) == -1
    p.mode = L.YP_MODE_YOLOV3
    heads[0].num_anchors = 9
    assert lib.yp_detect_yolov3(heads, 1, ctypes.byref(p), fake, 10, fake, None, fake, 1 << 30, None) == -3
    heads[0].num_anchors = 3
    p.batch = 0
    assert lib.yp_detect_yolov3(heads, 1, ctypes.byref(p), fake, 10, fake, None, fake, 1 << 30, None) == -1
    assert lib.yp_bbox_iou(None, fake, 3, 4, fake, None) == -1
    assert lib.yp_bbox_iou(fake, fake, 3, 2, fake, None) == -1
    assert lib.yp_write_results(None, 10, ctypes.byref(p), fake, 10, fake, None, fake, 1 << 30, None) == -1
    # unknown flag bits are refused (the field is no stage mask in the shipping library)
    p.batch, p.flags = 1, 4
    assert lib.yp_detect_yolov3(heads, 1, ctypes.byref(p), fake, 10, fake, None, fake, 1 << 30, None) == -1
    assert lib.yp_write_results(fake, 10, ctypes.byref(p), fake, 10, fake, None, fake, 1 << 30, None) == -1
    # the rescale flag needs a frame, and the head geometry (not offered on decoded rows)
    p.flags = L.YP_FLAG_RESCALE
    assert lib.yp_detect_yolov3(heads, 1, ctypes.byref(p), fake, 10, fake, None, fake, 1 << 30, None) == -1
    p.im_h, p.im_w = 1280.0, 1920.0
    assert lib.yp_write_results(fake, 10, ctypes.byref(p), fake, 10, fake, None, fake, 1 << 30, None) == -1
    p.im_h = p.im_w = 0.0
    p.flags = 0x100
    assert lib.yp_detect_yolov3(heads, 1, ctypes.byref(p), fake, 10, fake, None, fake, 1 << 30, None) == -1
    p.flags = 0
    # reference-mode entries pad their outputs to max_out rows per image: 1 <= max_out <= 2^20
    r = L.YpParams()
    r.mode, r.batch, r.num_classes, r.conf_thr, r.nms_thr = L.YP_MODE_REFERENCE, 1, 5, 0.6, 0.5
    r.im_h, r.im_w = 1280.0, 1920.0
    rhead = L.make_head(None, 13, 13, [(1, 1)] * 5)
    rhead.data = 0x1000
    for bad in (0, -1, (1 << 20) + 1):
        r.max_out = bad
        assert lib.yp_detect_reference(ctypes.byref(rhead), ctypes.byref(r), fake, fake, fake, fake, fake,
                                       fake, fake, 1 << 30, None) == -1
        assert lib.yp_filter_boxes_reference(fake, fake, fake, fake, 845, ctypes.byref(r), fake, fake,
                                             fake, fake, fake, fake, fake, 1 << 30, None) == -1


def test_multi_set_arguments_are_validated_without_a_gpu(pkg):
    L, lib = pkg._lib, pkg._lib.load()
    p = L.YpParams()
    p.mode, p.batch, p.num_classes, p.conf_thr, p.nms_thr = L.YP_MODE_YOLOV3, 2, 5, 0.5, 0.4
    fake = ctypes.c_void_p(0x1000)

    def sets(n, grid=13):
        hs = [L.make_head(None, 13, 13, [(1, 1)] * 3, 32, 32) for _ in range(n)]
        for h in hs:
            h.data = 0x1000
        hs[-1].grid_h = grid
        return (L.YpHead * n)(*hs)

    call = lambda heads, n, cap=10: lib.yp_detect_yolov3_multi(
        heads, 1, n, ctypes.byref(p), fake, cap, fake, None, fake, 1 << 30, None)
    assert call(sets(2), 0) == -3 and call(sets(2), L.YP_MAX_SETS + 1) == -3
    assert call(sets(2, grid=14), 2) == -1         # a set whose geometry differs from set 0
    hs = sets(2)
    hs[1].data = 0
    assert call(hs, 2) == -1                       # null data pointer in a later set
    # a workspace sized for one set is too small for two
    one = lib.yp_detect_yolov3_workspace_bytes(sets(1), 1, 2, 5)
    two = lib.yp_detect_yolov3_workspace_bytes(sets(1), 1, 4, 5)
    assert two > one
    assert lib.yp_detect_yolov3_multi(sets(2), 1, 2, ctypes.byref(p), fake, 10, fake, None, fake, one,
                                      None) == -2


def test_product_refuses_to_run_without_a_device(pkg):
    """No CPU fallback: CPU tensors are rejected (or, with no GPU at all, the call raises)."""
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the refusal path for missing devices is not reachable")
    with pytest.raises(Exception):
        pkg.util.write_results(torch.zeros(1, 10, 10), 0.5, 5)
    with pytest.raises(Exception):
        pkg.DetectNet.predict_transform(torch.zeros(1, 8450))
    with pytest.raises(Exception):
        pkg.util.YoloV3Post(1, 416)


def test_product_never_imports_the_oracle():
    """The package, the C sources and the diagnostic tools stay clear of oracle/ (only tests,
    the smoke check and bench.py's CPU baseline legs may use it)."""
    here = os.path.dirname(__file__)
    for sub in ("2d_object_detection_amd", "tools", "include"):
        for dirpath, _, files in os.walk(os.path.join(here, "..", sub)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".sh")):
                    text = open(os.path.join(dirpath, f)).read()
                    assert "import oracle" not in text and "from oracle" not in text, f
                    assert "ypo_" not in text, f


def test_params_mirror_reference_names(pkg):
    P = pkg.params
    assert (P.grid_height, P.grid_width, P.num_anchors, P.num_classes, P.pred_vec_len) == (13, 13, 5, 5, 10)
    assert (P.object_conf, P.nms_conf, P.max_boxes, P.im_height, P.im_width) == (.6, .5, 20, 1280, 1920)
    a = pkg.DetectNet.get_anchors()
    assert tuple(a.shape) == (5, 2) and a.dtype == torch.float32
    assert a.numpy().shape == (5, 2)
    assert pkg.DetectNet.get_new_dim(416, 3, 1, 1) == 416


def test_params_equal_the_references_own_constants(pkg):
    """Every constant this package shares with the reference's params.py has the reference's value
    and type of value: compared with tests/golden/reference_params.json, which
    tests/golden/make_params_fixture.py wrote by loading the reference's module itself."""
    import json
    import numpy as np
    here = os.path.dirname(os.path.abspath(__file__))
    ref = json.load(open(os.path.join(here, "golden", "reference_params.json")))
    P = pkg.params
    shared = [k for k in ref if hasattr(P, k)]
    # everything the post-processing path, the grid codec and the losses read
    for needed in ("object_conf", "nms_conf", "max_boxes", "num_anchors", "num_classes", "pred_vec_len",
                   "true_vec_len", "grid_height", "grid_width", "im_height", "im_width", "YOLO_ANCHORS",
                   "coord_loss_weight", "obj_loss_weight", "noobj_loss_weight", "scaled_height", "scaled_width",
                   "grid_stride"):
        assert needed in shared, needed
    for k in shared:
        got = getattr(P, k)
        if k == "YOLO_ANCHORS":
            assert isinstance(got, np.ndarray) and got.dtype == np.float64        # a float64 literal, as there
            assert got.tolist() == ref[k]
        else:
            assert got == ref[k] and type(got) is type(ref[k]), k


def test_abi_validation_layer_under_host_sanitizers():
    """The argument validation and workspace layout code of the C ABI, compiled for the host with
    AddressSanitizer + UndefinedBehaviorSanitizer (tests/abi_sanitized.cpp linked against the
    library's own sources) and run without a GPU: every call is rejected before a launch, and the
    sanitizers must stay silent."""
    import shutil
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc is not installed here")
    csrc = os.path.join(here, "..", "2d_object_detection_amd", "csrc")
    build = os.path.join(here, "_build")
    os.makedirs(build, exist_ok=True)
    exe = os.path.join(build, "abi_sanitized")
    srcs = [os.path.join(here, "abi_sanitized.cpp")] + [os.path.join(csrc, f) for f in
                                                         ("yolopost.hip", "gridcodec.hip", "decode_grad.hip", "metrics.hip")]
    deps = srcs + [os.path.join(csrc, f) for f in ("yp_graph.h", "yp_math.h", "yp_tile.h")] + \
        [os.path.join(here, "..", "include", "yolopost.h")]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(f) for f in deps):
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-ffp-contract=off",
                               "-fhip-fp32-correctly-rounded-divide-sqrt",
                               "-I" + os.path.join(here, "..", "include"), "-I" + csrc,
                               "-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fsanitize=undefined",
                               "-Xarch_host", "-fno-sanitize-recover=undefined", "-o", exe] + srcs)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", HIP_VISIBLE_DEVICES="")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.strip().splitlines()[-1].startswith("ok "), out.stdout[-500:]
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-4000:]


def test_flag_policy_of_the_python_engines(pkg):
    """Which YpParams.flags the host layer sets (pure host logic, no device needed)."""
    util, L = pkg.util, pkg._lib
    # the large-segment path: explicit choice wins, otherwise low thresholds on many anchors
    assert util._dense_nms(None, 0.001, 151200, 5) is True
    assert util._dense_nms(None, 0.001, 10647, 5) is True
    assert util._dense_nms(None, 0.001, 10647, 80) is False       # few anchors per class
    assert util._dense_nms(None, 0.5, 151200, 5) is True           # 1920x1280: thousands per segment at any threshold
    assert util._dense_nms(None, 0.5, 22743, 5) is False and util._dense_nms(None, 0.5, 10647, 5) is False
    assert util._dense_nms(False, 0.001, 151200, 5) is False and util._dense_nms(True, 0.9, 10, 5) is True
    p = util._v3_params(4, 5, 0.5, 0.4)
    assert (p.mode, p.batch, p.num_classes, p.max_out, p.flags) == (L.YP_MODE_YOLOV3, 4, 5, 0, 0)
    p = util._v3_params(4, 5, 0.5, 0.4, batch_offset=8, dense=True, frame=(1280, 1920))
    assert p.flags == L.YP_FLAG_DENSE_NMS | L.YP_FLAG_RESCALE and p.batch_offset == 8
    assert (p.im_h, p.im_w) == (1280.0, 1920.0)
    # and the library accepts exactly that combination (rejected here only for its fake pointers' sizes)
    lib = L.load()
    fake = ctypes.c_void_p(0x1000)
    heads = (L.YpHead * 1)(L.make_head(None, 13, 13, [(1, 1)] * 3, 32, 32))
    heads[0].data = 0x1000
    assert lib.yp_detect_yolov3(heads, 1, ctypes.byref(p), fake, 10, fake, None, fake, 16, None) == -2   # workspace, not flags
