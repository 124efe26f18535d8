"""DetectNet post-processing with the reference's call surface (detect_net.py:75-183).

``DetectNet.predict_transform`` / ``DetectNet.filter_boxes`` / ``DetectNet.get_anchors``
keep the reference's signatures, argument meaning and return structure so the caller in
train.py:347-349 (``run_validation``) works unchanged; the work is done by the HIP kernels
of libyolopost.so on the current ROCm device.  Tensors are ``torch.Tensor`` where the
reference used eager TF tensors.  ``DetectNet.detect`` is the fused composition
``filter_boxes(predict_transform(x))`` in one pass (same results, no intermediate tensors).

All configuration is read from ``params`` at call time, as in the reference.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from . import params


def _device_tensor(x, name):
    """Inputs may arrive as numpy arrays or CPU tensors; the computation is always on the GPU."""
    t = x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))
    if not t.is_cuda:
        if not torch.cuda.is_available():
            raise _lib.YoloPostError(
                "%s: no ROCm device available; this library has no CPU path" % name)
        t = t.cuda()
    return _lib.as_f32(t, name)


def _ref_params(batch):
    p = _lib.YpParams()
    p.mode = _lib.YP_MODE_REFERENCE
    p.batch = int(batch)
    p.num_classes = int(params.num_classes)
    p.max_out = int(params.max_boxes)
    p.conf_thr = float(params.object_conf)
    p.nms_thr = float(params.nms_conf)
    p.im_h = float(params.im_height)
    p.im_w = float(params.im_width)
    return p


def _pack(per_image):
    """np.array(list_of_per_image_arrays) as the reference does (detect_net.py:146): a
    regular array when every image has the same count (so ``.size == 0`` exactly when
    every image is empty, the test train.py:349 makes), otherwise an object array."""
    shapes = {a.shape for a in per_image}
    if len(shapes) == 1:
        return np.stack(per_image) if per_image else np.zeros((0,))
    out = np.empty(len(per_image), dtype=object)
    for i, a in enumerate(per_image):
        out[i] = a
    return out


def _unpad(outs):
    """Padded device outputs -> the reference's four numpy arrays.  The six outputs are views of
    one device buffer (see _alloc_outputs), so this is ONE device-to-host copy."""
    B, mb = outs[0].shape[0], outs[0].shape[1]
    n = B * mb
    n4 = (n + 3) & ~3                                # as laid out by _alloc_outputs
    host = outs[0]._yp_buffer.cpu().numpy()          # int32 words
    boxes = host[0:4 * n].view(np.float32).reshape(B, mb, 4)
    o = 4 * n
    scores = host[o:o + n].view(np.float32).reshape(B, mb)
    classes = host[o + n4:o + n4 + n].reshape(B, mb)
    index = host[o + 2 * n4:o + 2 * n4 + n].reshape(B, mb)
    counts = host[o + 3 * n4:o + 3 * n4 + B]
    pb = [boxes[b, :counts[b]].copy() for b in range(B)]
    ps = [scores[b, :counts[b]].copy() for b in range(B)]
    pc = [classes[b, :counts[b]].astype(np.int64) for b in range(B)]   # K.argmax is int64
    pi = [index[b, :counts[b]].copy() for b in range(B)]
    return _pack(pb), _pack(ps), _pack(pc), _pack(pi)


def _alloc_outputs(B, device):
    """(boxes (B,mb,4) f32, scores (B,mb) f32, classes (B,mb) i32, index (B,mb) i32, counts (B) i32,
    survivor counts (B) i32) as views of one int32 buffer: one allocation, and one copy when
    the results go to the host.  mb is padded to a multiple of 4 rows' worth of words so that
    every view starts 16-byte aligned."""
    mb = int(params.max_boxes)
    n = B * mb
    n4 = (n + 3) & ~3                      # words per (B, mb) plane, 16-byte multiple
    words = 4 * n + 3 * n4 + 2 * ((B + 3) & ~3)
    buf = torch.empty((words,), dtype=torch.int32, device=device)
    o = 0
    boxes = buf[o:o + 4 * n].view(torch.float32).view(B, mb, 4); o += 4 * n
    scores = buf[o:o + n].view(torch.float32).view(B, mb); o += n4
    classes = buf[o:o + n].view(B, mb); o += n4
    index = buf[o:o + n].view(B, mb); o += n4
    counts = buf[o:o + B]; o += (B + 3) & ~3
    surv = buf[o:o + B]
    boxes._yp_buffer = buf
    return boxes, scores, classes, index, counts, surv


def _anchor_array(A):
    return np.ascontiguousarray(np.asarray(params.YOLO_ANCHORS, dtype=np.float32).reshape(A, 2))


def _decode_forward(pred, H, W, A, C):
    B, dev = pred.shape[0], pred.device
    center = torch.empty((B, H, W, A, 2), dtype=torch.float32, device=dev)
    wh = torch.empty((B, H, W, A, 2), dtype=torch.float32, device=dev)
    obj = torch.empty((B, H, W, A, 1), dtype=torch.float32, device=dev)
    cls = torch.empty((B, H, W, A, C), dtype=torch.float32, device=dev)
    if B == 0:
        return center, wh, obj, cls
    anc = _anchor_array(A)
    with _lib.on_device(dev):
        rc = _lib.load().yp_predict_transform_reference(
            pred.data_ptr(), B, H, W, A, C,
            anc.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
            center.data_ptr(), wh.data_ptr(), obj.data_ptr(), cls.data_ptr(),
            _lib.stream_ptr(dev))
    _lib.check(rc, "yp_predict_transform_reference")
    return center, wh, obj, cls


class _PredictTransform(torch.autograd.Function):
    """predict_transform with a hand-written backward kernel, so that the training losses
    (train.py:123-197) can differentiate through it as tf.GradientTape does in the
    reference (train.py:264-267)."""

    @staticmethod
    def forward(ctx, pred, H, W, A, C):
        ctx.save_for_backward(pred)
        ctx.geom = (H, W, A, C)
        return _decode_forward(pred, H, W, A, C)

    @staticmethod
    def backward(ctx, g_center, g_wh, g_obj, g_cls):
        (pred,) = ctx.saved_tensors
        H, W, A, C = ctx.geom
        B, dev = pred.shape[0], pred.device
        grad = torch.empty_like(pred)
        if B == 0:
            return grad, None, None, None, None
        gs = [None if g is None else _lib.as_f32(g, "upstream gradient")
              for g in (g_center, g_wh, g_obj, g_cls)]
        anc = _anchor_array(A)
        with _lib.on_device(dev):
            rc = _lib.load().yp_predict_transform_reference_backward(
                pred.data_ptr(), B, H, W, A, C,
                anc.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                *[None if g is None else g.data_ptr() for g in gs],
                grad.data_ptr(), _lib.stream_ptr(dev))
        _lib.check(rc, "yp_predict_transform_reference_backward")
        return grad, None, None, None, None


class DetectNet():
    # survivors of the confidence filter per image for the most recent filter_boxes /
    # detect call (device tensor); diagnostic only
    last_survivor_counts = None

    @staticmethod
    def get_new_dim(input_dim, kern_sz, pdng, strd):
        """Convolution output size (detect_net.py:76-79)."""
        return int((input_dim - kern_sz + 2 * pdng) / strd) + 1

    @staticmethod
    def get_anchors():
        """params.YOLO_ANCHORS as a float32 (A, 2) tensor (detect_net.py:84-87).  It lives
        on the host so that ``.numpy()`` and iteration work as create_dataset.py:67,79 do."""
        return torch.tensor(np.asarray(params.YOLO_ANCHORS), dtype=torch.float32)

    @staticmethod
    def predict_transform(predictions):
        """Raw head -> (center_coords, wh_coords, obj_scores, class_probs)
        (detect_net.py:152-183).  Accepts anything reshapeable to
        [-1, grid_height, grid_width, num_anchors, pred_vec_len].  When the input requires
        grad the four outputs are differentiable (backward runs a HIP kernel too)."""
        H, W = int(params.grid_height), int(params.grid_width)
        A, C = int(params.num_anchors), int(params.num_classes)
        V = int(params.pred_vec_len)
        if V != 5 + C:
            raise ValueError("params.pred_vec_len must equal 5 + params.num_classes")
        pred = _device_tensor(predictions, "predictions").reshape(-1, H, W, A, V)
        if torch.is_grad_enabled() and pred.requires_grad:
            return _PredictTransform.apply(pred, H, W, A, C)
        return _decode_forward(pred, H, W, A, C)

    @staticmethod
    def dense_relu_predict_transform(features, weight, bias, return_head=False):
        """The head's last layer and the decode in ONE kernel: relu(features @ weight + bias)
        (detect_net.py:67-68) decoded as predict_transform does (detect_net.py:152-183) from the
        registers it was accumulated in, without storing the raw head.  ``features`` (B, F),
        ``weight`` (F, H*W*A*(5+C)), ``bias`` (H*W*A*(5+C)); params.num_classes must be 5.  The
        four outputs equal predict_transform of the stored head bit for bit (``return_head``
        adds that head as a fifth result).  The dot products are plain loops: this shows the
        epilogue hook a producer kernel can call, it is not a tuned dense layer."""
        H, W = int(params.grid_height), int(params.grid_width)
        A, C = int(params.num_anchors), int(params.num_classes)
        x = _device_tensor(features, "features")
        dev = x.device
        w = _device_tensor(weight, "weight").to(dev)
        bvec = _device_tensor(bias, "bias").to(dev)
        n_out = H * W * A * (5 + C)
        if x.dim() != 2 or w.dim() != 2 or w.shape != (x.shape[1], n_out) or bvec.numel() != n_out:
            raise ValueError("features (B, F), weight (F, %d), bias (%d) expected" % (n_out, n_out))
        B, F = x.shape
        center = torch.empty((B, H, W, A, 2), dtype=torch.float32, device=dev)
        wh = torch.empty((B, H, W, A, 2), dtype=torch.float32, device=dev)
        obj = torch.empty((B, H, W, A, 1), dtype=torch.float32, device=dev)
        cls = torch.empty((B, H, W, A, C), dtype=torch.float32, device=dev)
        head = torch.empty((B, n_out), dtype=torch.float32, device=dev) if return_head else None
        if B > 0:
            anc = _anchor_array(A)
            with _lib.on_device(dev):
                rc = _lib.load().yp_dense_relu_predict_transform_reference(
                    x.data_ptr(), w.data_ptr(), bvec.data_ptr(), B, F, H, W, A, C,
                    anc.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                    head.data_ptr() if return_head else None,
                    center.data_ptr(), wh.data_ptr(), obj.data_ptr(), cls.data_ptr(), _lib.stream_ptr(dev))
            _lib.check(rc, "yp_dense_relu_predict_transform_reference")
        return (center, wh, obj, cls, head) if return_head else (center, wh, obj, cls)

    @staticmethod
    def filter_boxes(transformed_predictions):
        """(center, wh, obj, class_probs) -> (boxes, scores, classes, nms_indexes) as numpy
        arrays of per-image arrays (detect_net.py:90-146)."""
        center, wh, obj, cls = transformed_predictions
        center = _device_tensor(center, "center_coords")
        dev = center.device
        wh = _device_tensor(wh, "wh_coords").to(dev)
        obj = _device_tensor(obj, "obj_scores").to(dev)
        cls = _device_tensor(cls, "class_probs").to(dev)
        B = center.shape[0]
        C = cls.shape[-1]
        if C != int(params.num_classes):
            raise ValueError("class_probs has %d classes, params.num_classes is %d"
                             % (C, params.num_classes))
        if B == 0:
            e = np.zeros((0,))
            return e, e, e, e
        N = obj.numel() // B
        if center.numel() != B * N * 2 or wh.numel() != B * N * 2 or cls.numel() != B * N * C:
            raise ValueError("inconsistent shapes in transformed_predictions")
        if params.print_max_score:                                   # detect_net.py:110
            print('max score in batch', float((obj * cls).max().item()))
        p = _ref_params(B)
        L = _lib.load()
        outs = _alloc_outputs(B, dev)
        ws = _lib.workspace(L.yp_detect_reference_workspace_bytes(B, N, 1, 1), dev)
        with _lib.on_device(dev):
            rc = L.yp_filter_boxes_reference(
                center.data_ptr(), wh.data_ptr(), obj.data_ptr(), cls.data_ptr(), N,
                ctypes.byref(p), *[o.data_ptr() for o in outs], ws.data_ptr(), ws.numel(),
                _lib.stream_ptr(dev))
        _lib.check(rc, "yp_filter_boxes_reference")
        DetectNet.last_survivor_counts = outs[5]
        return _unpad(outs)

    @staticmethod
    def detect_device(predictions):
        """Fused predict_transform + filter_boxes from the raw head, results left on the
        device: (boxes (B,max_boxes,4), scores, classes int32, nms_index int32, counts (B),
        survivor_counts (B)); rows beyond counts[b] are unspecified.  No host sync."""
        H, W = int(params.grid_height), int(params.grid_width)
        A, C = int(params.num_anchors), int(params.num_classes)
        pred = _device_tensor(predictions, "predictions").reshape(-1, H * W * A * (5 + C))
        B, dev = pred.shape[0], pred.device
        if B == 0:
            raise ValueError("empty batch")
        anc = np.asarray(params.YOLO_ANCHORS, dtype=np.float32).reshape(A, 2)
        head = _lib.make_head(pred, H, W, [tuple(a) for a in anc])
        p = _ref_params(B)
        L = _lib.load()
        outs = _alloc_outputs(B, dev)
        ws = _lib.workspace(L.yp_detect_reference_workspace_bytes(B, H, W, A), dev)
        with _lib.on_device(dev):
            rc = L.yp_detect_reference(ctypes.byref(head), ctypes.byref(p),
                                       *[o.data_ptr() for o in outs], ws.data_ptr(), ws.numel(),
                                       _lib.stream_ptr(dev))
        _lib.check(rc, "yp_detect_reference")
        DetectNet.last_survivor_counts = outs[5]
        return outs

    @staticmethod
    def detect(predictions):
        """filter_boxes(predict_transform(predictions)) in one fused pass, returned in the
        reference's numpy structure (what train.py:347-348 computes)."""
        return _unpad(DetectNet.detect_device(predictions))
