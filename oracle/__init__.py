"""CPU oracle for the YOLO post-processing path -- TEST INFRASTRUCTURE ONLY.

This package wraps ``oracle/ypo.c`` (a single-threaded C restatement of the reference
algorithm, see the header of that file for the file:line citations) behind numpy
arrays.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it; the product package never does.

PARITY UNPINNED: the reference ships no tests or golden vectors for this path and cannot
be imported here (it needs TensorFlow).  The oracle is pinned by hand-derivable known
answers, by the independent numpy restatement in ``oracle/np_oracle.py`` and by the
reference's own encode formula (create_dataset.py:101-110) used as a round trip; the
restatement of tf.image.non_max_suppression (the third-party routine behind detect_net.py:136)
also reproduces the cases of TensorFlow's own unit test of the op (tests/tf_nms_vectors.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libypo.so")

LIBM = 0   # exp flavour: libm expf
SPEC = 1   # exp flavour: fully specified binary32 sequence (bit-reproducible)

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)


def build(force=False):
    """Compile oracle/ypo.c with gcc into oracle/_build/libypo.so."""
    src = os.path.join(_HERE, "ypo.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.ypo_exp_spec.restype = ctypes.c_float
        L.ypo_exp_spec.argtypes = [ctypes.c_float]
        L.ypo_sigmoid.restype = ctypes.c_float
        L.ypo_sigmoid.argtypes = [ctypes.c_float, ctypes.c_int]
        L.ypo_tf_iou.restype = ctypes.c_float
        L.ypo_tf_iou.argtypes = [_f32p, _f32p]
        L.ypo_n_iou.restype = ctypes.c_float
        L.ypo_n_iou.argtypes = [_f32p, _f32p]
        L.ypo_tf_nms.restype = ctypes.c_int
        L.ypo_n_write_results.restype = ctypes.c_long
        L.ypo_n_detect.restype = ctypes.c_long
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t=_f32p):
    return a.ctypes.data_as(t)


# ----------------------------------------------------------------------------- scalars

def exp_spec(x):
    x = _f32(x)
    out = np.empty_like(x)
    L = lib()
    flat_in, flat_out = x.reshape(-1), out.reshape(-1)
    for i in range(flat_in.size):
        flat_out[i] = L.ypo_exp_spec(float(flat_in[i]))
    return out


def sigmoid(x, flavour=SPEC):
    x = _f32(x)
    out = np.empty_like(x)
    L = lib()
    flat_in, flat_out = x.reshape(-1), out.reshape(-1)
    for i in range(flat_in.size):
        flat_out[i] = L.ypo_sigmoid(float(flat_in[i]), flavour)
    return out


def tf_iou(bi, bj):
    bi, bj = _f32(bi), _f32(bj)
    return float(lib().ypo_tf_iou(_p(bi), _p(bj)))


def n_iou(b1, b2):
    b1, b2 = _f32(b1), _f32(b2)
    return float(lib().ypo_n_iou(_p(b1), _p(b2)))


# ------------------------------------------------------------------------------ Mode R

def r_predict_transform(pred, H, W, A, C, anchors, flavour=SPEC):
    """detect_net.py:152-183.  pred (B, H*W*A*(5+C)) -> center, wh, obj, cls."""
    pred = _f32(pred)
    B = pred.size // (H * W * A * (5 + C))
    anchors = _f32(anchors).reshape(A, 2)
    center = np.empty((B, H, W, A, 2), np.float32)
    wh = np.empty((B, H, W, A, 2), np.float32)
    obj = np.empty((B, H, W, A, 1), np.float32)
    cls = np.empty((B, H, W, A, C), np.float32)
    lib().ypo_r_predict_transform(_p(pred), B, H, W, A, C, _p(anchors), flavour,
                                  _p(center), _p(wh), _p(obj), _p(cls))
    return center, wh, obj, cls


def tf_nms(boxes, scores, max_out, iou_thr):
    boxes, scores = _f32(boxes).reshape(-1, 4), _f32(scores).reshape(-1)
    K = scores.size
    out = np.empty(max(max_out, 1), np.int32)
    n = lib().ypo_tf_nms(_p(boxes), _p(scores), K, int(max_out), ctypes.c_float(iou_thr),
                         _p(out, _i32p))
    return out[:n].copy()


def r_filter_boxes(center, wh, obj, cls, conf_thr, nms_thr, max_boxes, im_h, im_w):
    """detect_net.py:90-146 on the decoded 4-tuple.

    Returns per-image lists (boxes[M,4] y1x1y2x2 pixels, scores[M], classes[M] int64,
    nms_index[M] int32) and the survivor count per image."""
    center, wh, obj, cls = _f32(center), _f32(wh), _f32(obj), _f32(cls)
    B = center.shape[0]
    C = cls.shape[-1]
    N = obj.size // B
    mb = max(int(max_boxes), 1)
    ob = np.zeros((B, mb, 4), np.float32)
    os_ = np.zeros((B, mb), np.float32)
    oc = np.zeros((B, mb), np.int64)
    oi = np.zeros((B, mb), np.int32)
    cnt = np.zeros(B, np.int32)
    surv = np.zeros(B, np.int32)
    lib().ypo_r_filter_boxes(_p(center), _p(wh), _p(obj), _p(cls), B, N, C,
                             ctypes.c_float(conf_thr), ctypes.c_float(nms_thr),
                             int(max_boxes), ctypes.c_float(im_h), ctypes.c_float(im_w),
                             _p(ob), _p(os_), _p(oc, _i64p), _p(oi, _i32p),
                             _p(cnt, _i32p), _p(surv, _i32p))
    boxes = [ob[b, :cnt[b]].copy() for b in range(B)]
    scores = [os_[b, :cnt[b]].copy() for b in range(B)]
    classes = [oc[b, :cnt[b]].copy() for b in range(B)]
    index = [oi[b, :cnt[b]].copy() for b in range(B)]
    return boxes, scores, classes, index, surv


def r_detect(pred, H, W, A, C, anchors, conf_thr, nms_thr, max_boxes, im_h, im_w,
             flavour=SPEC):
    center, wh, obj, cls = r_predict_transform(pred, H, W, A, C, anchors, flavour)
    return r_filter_boxes(center, wh, obj, cls, conf_thr, nms_thr, max_boxes, im_h, im_w)


# ------------------------------------------------------------------------------ Mode N

def scaled_anchors(anchors_px, stride_x, stride_y):
    """float32(anchor_px / stride), the division done in double as the tutorial does."""
    a = np.asarray(anchors_px, dtype=np.float64).reshape(-1, 2)
    return np.stack([a[:, 0] / float(stride_x), a[:, 1] / float(stride_y)], 1).astype(np.float32)


def n_predict_transform(head, inp_dim, anchors_px, num_classes, flavour=SPEC):
    """Tutorial predict_transform for one NCHW head (B, A*(5+C), gh, gw).

    inp_dim is an int (square) or (H, W).  Returns rows (B, gh*gw*A, 5+C)."""
    head = _f32(head)
    B, ch, gh, gw = head.shape
    A = len(anchors_px)
    C = num_classes
    assert ch == A * (5 + C)
    in_h, in_w = (inp_dim, inp_dim) if np.isscalar(inp_dim) else inp_dim
    sx, sy = in_w // gw, in_h // gh
    anc = scaled_anchors(anchors_px, sx, sy)
    rows = np.empty((B, gh * gw * A, 5 + C), np.float32)
    lib().ypo_n_predict_transform(_p(head), B, A, C, gh, gw, ctypes.c_float(sx),
                                  ctypes.c_float(sy), _p(anc), flavour, _p(rows))
    return rows


def n_write_results(rows, confidence, num_classes, nms_conf=0.4, cap=None):
    """Tutorial write_results.  Returns (out[D,8], survivors_per_image, iou_pairs)."""
    rows = _f32(rows)
    B, N, V = rows.shape
    assert V == 5 + num_classes
    cap = B * N if cap is None else cap
    out = np.zeros((max(cap, 1), 8), np.float32)
    surv = np.zeros(B, np.int32)
    pairs = np.zeros(1, np.int64)
    D = lib().ypo_n_write_results(_p(rows), B, N, num_classes, ctypes.c_float(confidence),
                                  ctypes.c_float(nms_conf), _p(out), ctypes.c_long(cap),
                                  _p(surv, _i32p), _p(pairs, _i64p))
    return out[:min(D, cap)].copy(), surv, int(pairs[0])


def n_detect(heads, inp_dim, anchors_px_per_head, num_classes, confidence, nms_conf=0.4,
             flavour=SPEC):
    """Decode all heads (head order kept) + write_results.  Returns (out, surv, pairs)."""
    rows = np.concatenate([n_predict_transform(h, inp_dim, a, num_classes, flavour)
                           for h, a in zip(heads, anchors_px_per_head)], axis=1)
    return n_write_results(rows, confidence, num_classes, nms_conf)


def n_bbox_iou(box1, boxes2):
    box1 = _f32(box1).reshape(-1)
    boxes2 = _f32(boxes2)
    M, stride = boxes2.shape
    out = np.empty(M, np.float32)
    lib().ypo_n_bbox_iou(_p(box1), _p(boxes2), M, stride, _p(out))
    return out
