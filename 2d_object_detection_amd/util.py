"""YOLOv3 three-stride post-processing: ``predict_transform``, ``write_results``,
``bbox_iou`` with the signatures of the YOLOv3 tutorial the reference credits
(README.md:1), plus ``YoloV3Post``, the fused raw-heads -> detections engine.

These functions are not part of the reference repository (its own path is the single-grid
``DetectNet`` in detect_net.py); they are the semantics the project brief names.  Tensors
are ``torch.Tensor`` on a ROCm device; all arithmetic runs in the HIP kernels of
libyolopost.so.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .synthetic import YOLOV3_ANCHORS


def _dims(inp_dim):
    if isinstance(inp_dim, (tuple, list)):
        return int(inp_dim[0]), int(inp_dim[1])          # (H, W)
    return int(inp_dim), int(inp_dim)


def _scaled_anchors(anchors, sx, sy):
    """float32(anchor_px / stride), the division carried out in double (the tutorial divides
    Python numbers and then builds a FloatTensor)."""
    a = np.asarray(anchors, dtype=np.float64).reshape(-1, 2)
    return [(float(np.float32(w / sx)), float(np.float32(h / sy))) for w, h in a]


def _head_desc(t, inp_dim, anchors, num_classes):
    in_h, in_w = _dims(inp_dim)
    if t.dim() != 4:
        raise ValueError("a detection head must be (B, A*(5+C), grid_h, grid_w)")
    B, ch, gh, gw = t.shape
    A = len(anchors)
    if ch != A * (5 + num_classes):
        raise ValueError("head has %d channels, expected %d anchors x (5 + %d classes)"
                         % (ch, A, num_classes))
    sx, sy = in_w // gw, in_h // gh
    if sx <= 0 or sy <= 0:
        raise ValueError("grid larger than the input size")
    return _lib.make_head(t, gh, gw, _scaled_anchors(anchors, sx, sy), sx, sy)


def predict_transform(prediction, inp_dim, anchors, num_classes, CUDA=True):
    """One NCHW detection head (B, A*(5+C), G, G) -> decoded rows (B, G*G*A, 5+C):
    [cx, cy, w, h, objectness, class scores] in input pixels.  ``inp_dim`` is an int or
    (H, W); ``anchors`` are pixel (w, h) pairs.  The input is not modified."""
    t = _lib.as_f32(prediction, "prediction")
    return predict_transform_heads([t], inp_dim, [anchors], num_classes)


def predict_transform_heads(heads, inp_dim, anchors_per_head, num_classes):
    """All heads in ONE launch; equals torch.cat of the per-head results along dim 1."""
    heads = [_lib.as_f32(h, "heads[%d]" % i) for i, h in enumerate(heads)]
    if not 1 <= len(heads) <= _lib.YP_MAX_HEADS:
        raise ValueError("1..%d heads supported" % _lib.YP_MAX_HEADS)
    dev = heads[0].device
    B = heads[0].shape[0]
    for i, h in enumerate(heads):
        if h.dim() != 4 or h.shape[0] != B:
            raise ValueError("heads[%d] has shape %r: every head must be (B, A*(5+C), gh, gw) "
                             "with the same batch B = %d" % (i, tuple(h.shape), B))
        if h.device != dev:
            raise ValueError("heads[%d] is on %s, heads[0] on %s" % (i, h.device, dev))
    descs = (_lib.YpHead * len(heads))(
        *[_head_desc(h, inp_dim, a, num_classes) for h, a in zip(heads, anchors_per_head)])
    N = sum(d.grid_h * d.grid_w * d.num_anchors for d in descs)
    rows = torch.empty((B, N, 5 + num_classes), dtype=torch.float32, device=dev)
    if B == 0:
        return rows
    with _lib.on_device(dev):
        rc = _lib.load().yp_predict_transform_yolov3(descs, len(heads), B, num_classes,
                                                     rows.data_ptr(), _lib.stream_ptr(dev))
    _lib.check(rc, "yp_predict_transform_yolov3")
    return rows


def _dense_nms(dense_nms, confidence, rows_per_image, num_classes):
    """Whether to ask for the multi-workgroup path for large NMS segments (YP_FLAG_DENSE_NMS):
    an explicit True/False wins; None decides from the configuration -- a confidence threshold
    below 0.1 on more than 2048 anchors per class lets thousands of candidates through per
    (image, class), and so does any threshold on images of 65 536 anchors or more (1920x1280:
    2 000 candidates per segment at 0.5, where that path is a little ahead, and several thousand
    in crowded scenes, where it is 15-20 % ahead; its two extra launches cost about 4 us when no
    segment is large).  Results do not depend on it, only the time does."""
    if dense_nms is None:
        per_class = rows_per_image > 2048 * max(int(num_classes), 1)
        return (confidence < 0.1 and per_class) or (rows_per_image >= 65536 and per_class)
    return bool(dense_nms)


def _v3_params(batch, num_classes, confidence, nms_conf, batch_offset=0, dense=False, frame=None):
    """``frame`` = (height, width) of the original images: kept boxes are mapped back from the
    letterboxed network input to that frame (YP_FLAG_RESCALE)."""
    p = _lib.YpParams()
    p.mode = _lib.YP_MODE_YOLOV3
    p.batch = int(batch)
    p.num_classes = int(num_classes)
    p.max_out = 0
    p.conf_thr = float(confidence)
    p.nms_thr = float(nms_conf)
    p.batch_offset = int(batch_offset)
    p.flags = _lib.YP_FLAG_DENSE_NMS if dense else 0
    if frame is not None:
        p.flags |= _lib.YP_FLAG_RESCALE
        p.im_h, p.im_w = float(frame[0]), float(frame[1])
    return p


def write_results_device(prediction, confidence, num_classes, nms_conf=0.4, out_cap=None,
                         want_stats=False, dense_nms=None):
    """write_results without the host synchronisation: returns (rows (cap, 8), count (1,)
    int32, stats or None), all on the device; rows[:count] are the detections."""
    pred = _lib.as_f32(prediction, "prediction")
    if pred.dim() != 3 or pred.shape[2] != 5 + num_classes:
        raise ValueError("prediction must be (B, N, 5 + num_classes)")
    B, N, _ = pred.shape
    dev = pred.device
    cap = B * N if out_cap is None else int(out_cap)
    out = torch.empty((max(cap, 1), 8), dtype=torch.float32, device=dev)
    count = torch.zeros((1,), dtype=torch.int32, device=dev)
    stats = (torch.zeros((_lib.YP_STATS_PLANES, B, num_classes), dtype=torch.int32, device=dev)
             if want_stats else None)
    if B == 0 or N == 0:
        return out, count, stats
    L = _lib.load()
    p = _v3_params(B, num_classes, confidence, nms_conf,
                   dense=_dense_nms(dense_nms, confidence, N, num_classes))
    ws = _lib.workspace(L.yp_rows_workspace_bytes(B, N, num_classes), dev)
    with _lib.on_device(dev):
        rc = L.yp_write_results(pred.data_ptr(), N, ctypes.byref(p), out.data_ptr(), cap,
                                count.data_ptr(), stats.data_ptr() if want_stats else None,
                                ws.data_ptr(), ws.numel(), _lib.stream_ptr(dev))
    _lib.check(rc, "yp_write_results")
    return out, count, stats


def write_results(prediction, confidence, num_classes, nms_conf=0.4, dense_nms=None):
    """Decoded rows (B, N, 5+C) -> detections (D, 8)
    [image, x1, y1, x2, y2, objectness, class score, class id], ordered by image, class id,
    objectness descending; the int 0 when nothing survives (as the tutorial returns)."""
    out, count, _ = write_results_device(prediction, confidence, num_classes, nms_conf,
                                         dense_nms=dense_nms)
    d = int(count.item())
    if d == 0:
        return 0
    return out[:d]


def bbox_iou(box1, box2):
    """IoU (pixel-inclusive, +1 form) of box1 (1, >=4) with every row of box2 (M, >=4);
    corners are the first four columns (x1, y1, x2, y2).  Returns (M,)."""
    b1 = _lib.as_f32(box1, "box1").reshape(-1)
    b2 = _lib.as_f32(box2, "box2")
    if b2.dim() == 1:
        b2 = b2.unsqueeze(0)
    if b1.numel() < 4 or b2.shape[1] < 4:
        raise ValueError("boxes need at least four columns")
    M = b2.shape[0]
    out = torch.empty((M,), dtype=torch.float32, device=b2.device)
    if M == 0:
        return out
    with _lib.on_device(b2.device):
        rc = _lib.load().yp_bbox_iou(b1.data_ptr(), b2.data_ptr(), M, b2.shape[1], out.data_ptr(),
                                     _lib.stream_ptr(b2.device))
    _lib.check(rc, "yp_bbox_iou")
    return out


class YoloV3Post:
    """Fused raw-heads -> detections engine for a fixed configuration.

    Buffers (workspace, output rows, count, statistics) are allocated once; ``run`` only
    enqueues three kernels (five with ``dense_nms``) on the current stream and returns device tensors, so it can sit
    in a serving loop without host synchronisation.  ``batch_offset`` is added to the image
    index in column 0 (used when a batch is sharded over several GPUs).  ``frame`` = (height,
    width) of the original images maps the kept boxes back from the letterboxed network input
    to that frame, clamped to it (after NMS, as the YOLOv3 tutorial does); ``dense_nms`` see
    ``_dense_nms``."""

    def __init__(self, batch, inp_dim, num_classes=5, confidence=0.5, nms_conf=0.4,
                 anchors=YOLOV3_ANCHORS, strides=(32, 16, 8), device=None, out_cap=None,
                 batch_offset=0, out=None, dense_nms=None, frame=None, _allocate=True):
        self.device = torch.device(device if device is not None
                                   else "cuda:%d" % torch.cuda.current_device())
        if self.device.type != "cuda":
            raise _lib.YoloPostError("YoloV3Post needs a ROCm device; there is no CPU path")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.batch, self.inp_dim, self.num_classes = int(batch), inp_dim, int(num_classes)
        self.anchors, self.strides = anchors, tuple(strides)
        in_h, in_w = _dims(inp_dim)
        self.grids = [(in_h // s, in_w // s) for s in self.strides]
        self.rows_per_image = sum(gh * gw * len(a) for (gh, gw), a in zip(self.grids, anchors))
        self.dense_nms = _dense_nms(dense_nms, confidence, self.rows_per_image, num_classes)
        self.params = _v3_params(batch, num_classes, confidence, nms_conf, batch_offset,
                                 dense=self.dense_nms, frame=frame)
        self._lib = _lib.load()
        shape_only = (_lib.YpHead * len(self.grids))(*[
            _lib.make_head(None, gh, gw, _scaled_anchors(a, in_w // gw, in_h // gh),
                           in_w // gw, in_h // gh)
            for (gh, gw), a in zip(self.grids, anchors)])
        nbytes = self._lib.yp_detect_yolov3_workspace_bytes(shape_only, len(self.grids),
                                                            self.batch, self.num_classes)
        if nbytes == 0:
            raise _lib.YoloPostError("unsupported head geometry")
        self._desc_cache = {}
        if not _allocate:     # geometry, parameters and head validation only (YoloV3PostMulti)
            return
        self.workspace = _lib.workspace(nbytes, self.device)
        if out is not None:
            # caller-provided output rows (e.g. the payload view of a gather buffer)
            _lib.require_device(out, "out")
            if (out.dtype != torch.float32 or out.dim() != 2 or out.shape[1] != 8
                    or not out.is_contiguous() or out.data_ptr() % 16):
                raise ValueError("out must be a contiguous 16-byte aligned float32 (cap, 8) tensor")
            self.cap, self.out = out.shape[0], out
        else:
            self.cap = self.batch * self.rows_per_image if out_cap is None else int(out_cap)
            self.out = torch.empty((max(self.cap, 1), 8), dtype=torch.float32, device=self.device)
        self.count = torch.zeros((1,), dtype=torch.int32, device=self.device)
        self.stats = torch.zeros((_lib.YP_STATS_PLANES, self.batch, self.num_classes),
                                 dtype=torch.int32, device=self.device)
        # everything the per-batch call needs, resolved once
        self._detect = self._lib.yp_detect_yolov3
        self._params_ref = ctypes.byref(self.params)
        self._out_ptr, self._count_ptr = self.out.data_ptr(), self.count.data_ptr()
        self._stats_ptr = self.stats.data_ptr()
        self._ws_ptr, self._ws_bytes = self.workspace.data_ptr(), self.workspace.numel()

    def _descs(self, heads):
        if len(heads) != len(self.grids):
            raise ValueError("expected %d heads" % len(self.grids))
        hs = []
        for h, (gh, gw), a in zip(heads, self.grids, self.anchors):
            _lib.require_device(h, "head")
            if h.dtype != torch.float32 or not h.is_contiguous():
                raise ValueError("heads must be contiguous float32")
            if tuple(h.shape) != (self.batch, len(a) * (5 + self.num_classes), gh, gw):
                raise ValueError("head shape %r does not match the engine" % (tuple(h.shape),))
            if h.device != self.device:
                raise ValueError("head is on %s, the engine on %s" % (h.device, self.device))
            hs.append(_head_desc(h, self.inp_dim, a, self.num_classes))
        return (_lib.YpHead * len(hs))(*hs)

    def run(self, heads, stream=None):
        """Enqueue decode + filter + NMS; returns (rows (cap, 8), count (1,) int32) on the
        device.  rows[:count] are valid once the stream has executed.  ``stream`` is a
        torch.cuda.Stream (default: the current stream).  The validated head descriptors
        are cached per set of input buffers -- keyed on address, shape, strides, dtype and
        device of every head, so a view of a cached buffer with another shape (a final
        partial batch ``buf[:n]``) is validated afresh and rejected -- and a serving loop that
        reuses its input buffers pays one C call per batch."""
        key = tuple((h.data_ptr(), h.shape, h.stride(), h.dtype, h.device) for h in heads)
        descs = self._desc_cache.get(key)
        if descs is None:
            descs = self._descs(heads)
            if len(self._desc_cache) > 64:
                self._desc_cache.clear()
            self._desc_cache[key] = descs
        stream = _lib.stream_ptr(self.device) if stream is None else stream.cuda_stream
        with _lib.on_device(self.device):   # kernels go to the current device: make it the engine's
            rc = self._detect(descs, len(descs), self._params_ref, self._out_ptr, self.cap,
                              self._count_ptr, self._stats_ptr, self._ws_ptr, self._ws_bytes, stream)
        if rc != 0:
            _lib.check(rc, "yp_detect_yolov3")
        return self.out, self.count

    def detect(self, heads):
        """run() + one host read of the count: the (D, 8) detections (D may be 0)."""
        out, count = self.run(heads)
        d = min(int(count.item()), self.cap)
        return out[:d]

    def segment_stats(self):
        """(survivors, kept, iou_pairs) per (image, class) of the last run, int32 (B, C)."""
        s = self.stats.cpu().numpy()
        return s[0], s[1], s[2]

    def tie_stats(self):
        """Per (image, class) of the last run, int32 (B, C): how many of the evaluated IoU pairs
        lay within 2**-20 (relative) of the NMS threshold -- the IoU-at-threshold ties, which
        the kernel decides with the exact divide-and-compare."""
        return self.stats[3].cpu().numpy()


class YoloV3PostMulti:
    """``n_sets`` independent batches of one configuration per call: ONE launch per stage for
    all of them (yp_detect_yolov3_multi).  ``run`` takes a list of ``n_sets`` head lists and
    returns (rows (n_sets, cap, 8), counts (n_sets,) int32) on the device; set s's detections
    are rows[s, :counts[s]] and equal what ``YoloV3Post.run`` gives for that batch alone."""

    def __init__(self, n_sets, batch, inp_dim, num_classes=5, confidence=0.5, nms_conf=0.4,
                 anchors=YOLOV3_ANCHORS, strides=(32, 16, 8), device=None, out_cap=None,
                 batch_offset=0, out=None, dense_nms=None, frame=None):
        if not 1 <= int(n_sets) <= _lib.YP_MAX_SETS:
            raise ValueError("1..%d sets per call" % _lib.YP_MAX_SETS)
        self.n_sets = int(n_sets)
        # geometry, validation and parameters are those of a single-set engine (which allocates
        # no device memory of its own here)
        self._one = YoloV3Post(batch, inp_dim, num_classes, confidence, nms_conf, anchors, strides,
                               device, batch_offset=batch_offset, dense_nms=dense_nms, frame=frame,
                               _allocate=False)
        o = self._one
        self.device, self.batch, self.num_classes = o.device, o.batch, o.num_classes
        self.grids, self.rows_per_image, self.params = o.grids, o.rows_per_image, o.params
        self._lib = _lib.load()
        in_h, in_w = _dims(inp_dim)
        shape_only = (_lib.YpHead * len(self.grids))(*[
            _lib.make_head(None, gh, gw, _scaled_anchors(a, in_w // gw, in_h // gh),
                           in_w // gw, in_h // gh)
            for (gh, gw), a in zip(self.grids, anchors)])
        nbytes = self._lib.yp_detect_yolov3_workspace_bytes(shape_only, len(self.grids),
                                                            self.n_sets * self.batch, self.num_classes)
        if nbytes == 0:
            raise _lib.YoloPostError("unsupported head geometry")
        self.workspace = _lib.workspace(nbytes, self.device)
        if out is not None:
            # caller-provided output rows (e.g. the payload view of a gather buffer)
            _lib.require_device(out, "out")
            if (out.dtype != torch.float32 or out.dim() != 3 or out.shape[0] != self.n_sets
                    or out.shape[2] != 8 or not out.is_contiguous() or out.data_ptr() % 16):
                raise ValueError("out must be a contiguous 16-byte aligned float32 "
                                 "(n_sets, cap, 8) tensor")
            self.cap, self.out = out.shape[1], out
        else:
            self.cap = self.batch * self.rows_per_image if out_cap is None else int(out_cap)
            self.out = torch.empty((self.n_sets, max(self.cap, 1), 8), dtype=torch.float32,
                                   device=self.device)
        self.count = torch.zeros((self.n_sets,), dtype=torch.int32, device=self.device)
        self.stats = torch.zeros((self.n_sets, _lib.YP_STATS_PLANES, self.batch, self.num_classes),
                                 dtype=torch.int32, device=self.device)
        self._desc_cache = {}

    def _descs(self, sets, n):
        if not 1 <= n <= self.n_sets:
            raise ValueError("expected 1..%d sets of heads, got %d" % (self.n_sets, n))
        flat = []
        for heads in sets:
            flat.extend(self._one._descs(heads))     # validates every head of every set
        return (_lib.YpHead * len(flat))(*flat)

    def run(self, sets, stream=None):
        """sets: 1..n_sets lists of heads (fewer than n_sets uses the first len(sets) slices
        of the outputs).  Returns (rows, counts) on the device."""
        n = len(sets)
        key = tuple((h.data_ptr(), h.shape, h.stride(), h.dtype, h.device) for heads in sets
                    for h in heads)
        descs = self._desc_cache.get(key)
        if descs is None:
            descs = self._descs(sets, n)
            if len(self._desc_cache) > 64:
                self._desc_cache.clear()
            self._desc_cache[key] = descs
        stream_obj = stream
        stream = _lib.stream_ptr(self.device) if stream is None else stream.cuda_stream
        with _lib.on_device(self.device):
            rc = self._lib.yp_detect_yolov3_multi(
                descs, len(self.grids), n, ctypes.byref(self.params), self.out.data_ptr(), self.cap,
                self.count.data_ptr(), self.stats.data_ptr(), self.workspace.data_ptr(),
                self.workspace.numel(), stream)
        if rc != 0:
            _lib.check(rc, "yp_detect_yolov3_multi")
        if n < self.n_sets:      # the sets this call did not use read as empty, not as the last call's
            with torch.cuda.stream(stream_obj if stream_obj is not None
                                   else torch.cuda.current_stream(self.device)):
                self.count[n:].zero_()
        return self.out, self.count

    def detect(self, sets):
        """run() + one host read of the counts: a list of (D_s, 8) detections, one per set."""
        out, count = self.run(sets)
        c = count.cpu().tolist()
        return [out[s, :min(int(c[s]), self.cap)] for s in range(len(sets))]

    def segment_stats(self, n=None):
        """(survivors, kept, iou_pairs, ties), each int32 (n_sets, B, C), of the last run."""
        s = self.stats.cpu().numpy()
        return s[:, 0], s[:, 1], s[:, 2], s[:, 3]
