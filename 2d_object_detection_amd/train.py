"""The training-side callers of the detection path with the reference's names
(train.py:110-114 create_mask, :123-197 loss_custom, :200-270 loss_keras, :273-276 grad,
:331-340 get_metrics, :343-353 run_validation, :279-280 comp_nms_gt).

The arithmetic runs in HIP kernels of libyolopost.so: ``loss_custom`` is one fused kernel that
produces the loss and the gradient of the raw head together; ``DetectNet.predict_transform``
is differentiable for any other loss written on top of it.  The model is whatever callable the
caller brings (a ``torch.nn.Module`` here, a Keras model in the reference).
"""
import ctypes

import numpy as np
import torch

from . import _lib
from . import create_dataset
from . import params
from .detect_net import DetectNet, _anchor_array, _device_tensor, _unpad


def create_mask(true_box_grid):
    """1.0 where the grid is non-zero (train.py:110-114)."""
    g = true_box_grid if isinstance(true_box_grid, torch.Tensor) else torch.as_tensor(np.asarray(true_box_grid))
    return (g != 0).to(torch.float32)


class _LossCustom(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, true_grid, geom, weights):
        H, W, A, C = geom
        B, dev = pred.shape[0], pred.device
        L = _lib.load()
        loss = torch.empty((4,), dtype=torch.float32, device=dev)
        need_grad = ctx.needs_input_grad[0]
        grad = torch.empty_like(pred) if need_grad else None
        ws = _lib.workspace(L.yp_loss_custom_reference_workspace_bytes(B, H, W, A), dev)
        anc = _anchor_array(A)
        with _lib.on_device(dev):
            rc = L.yp_loss_custom_reference(
                pred.data_ptr(), true_grid.data_ptr(), B, H, W, A, C,
                anc.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), *[float(w) for w in weights],
                loss.data_ptr(), grad.data_ptr() if need_grad else None, ws.data_ptr(), ws.numel(),
                _lib.stream_ptr(dev))
        _lib.check(rc, "yp_loss_custom_reference")
        ctx.grad = grad
        total, parts = loss[0].clone(), loss[1:].clone()
        ctx.mark_non_differentiable(parts)
        return total, parts

    @staticmethod
    def backward(ctx, g_total, _g_parts):
        grad = ctx.grad
        return (None if grad is None else grad * g_total), None, None, None


def loss_from_logits(y_, true_box_grid, return_parts=False):
    """loss_custom on an already computed head output ``y_`` (train.py:128-197): a scalar device
    tensor, differentiable with respect to ``y_``.  With return_parts also the (confidence,
    classification, coordinates) sums of train.py:190-192."""
    H, W = int(params.grid_height), int(params.grid_width)
    A, C = int(params.num_anchors), int(params.num_classes)
    pred = _device_tensor(y_, "y_").reshape(-1, H * W * A * (5 + C))
    tg = _device_tensor(true_box_grid, "true_box_grid").to(pred.device).reshape(-1, H, W, A, 5)
    if tg.shape[0] != pred.shape[0] or pred.shape[0] == 0:
        raise ValueError("y_ and true_box_grid must hold the same, non-zero number of images")
    weights = (params.coord_loss_weight, params.obj_loss_weight, params.noobj_loss_weight,
               params.detection_iou)
    total, parts = _LossCustom.apply(pred, tg, (H, W, A, C), weights)
    return (total, parts) if return_parts else total


def loss_custom(x, true_box_grid, model, true_box_mask=None, training=True, count=-1):
    """train.py:123-197 with its signature: runs the model, then the fused loss kernel.
    ``true_box_mask`` and ``count`` only fed the reference's debug printing and are ignored."""
    if hasattr(model, "train"):
        model.train(bool(training))
    return loss_from_logits(model(x), true_box_grid)


def loss_keras(true_box_grid, y_):
    """train.py:200-270: the same loss on a precomputed head output, Keras argument order."""
    return loss_from_logits(y_, true_box_grid)


def grad(model, inputs, true_box_grid, box_mask=None, count=-1):
    """(loss, gradients of the model's trainable parameters) as train.py:273-276."""
    loss_value = loss_custom(inputs, true_box_grid, model, box_mask, True, count)
    weights = [p for p in model.parameters() if p.requires_grad]
    return loss_value, list(torch.autograd.grad(loss_value, weights, allow_unused=True))


def detection_rows(detections):
    """The padded device outputs of ``DetectNet.detect_device`` as the detection list the
    matching kernel reads: (rows (B*max_boxes, 8) float32 [image, x1, y1, x2, y2, score, score,
    class id], count (1,) int32), valid rows first in (image, NMS rank) order -- i.e. by
    descending score within an image.  All on the device, no host synchronisation."""
    boxes, scores, classes, _index, counts = detections[:5]
    B, mb = scores.shape
    dev = scores.device
    img = torch.arange(B, device=dev, dtype=torch.float32).view(B, 1).expand(B, mb)
    # the reference orders a box [y1, x1, y2, x2] (detect_net.py:97-102)
    rows = torch.stack([img, boxes[..., 1], boxes[..., 0], boxes[..., 3], boxes[..., 2], scores, scores,
                        classes.to(torch.float32)], dim=-1).reshape(B * mb, 8)
    valid = (torch.arange(mb, device=dev).view(1, mb) < counts.view(B, 1)).reshape(-1)
    order = torch.argsort((~valid).to(torch.int8), stable=True)
    return rows[order].contiguous(), valid.sum().to(torch.int32).reshape(1)


class MatchResult:
    """What ``comp_nms_gt`` leaves on the device: ``det_match`` (cap,) int32 -- the ground-truth
    row a true positive claims, or -1 none / -2 duplicate / -3 other class --, ``det_iou`` (cap,)
    float32 best IoU with any ground-truth box of the image, ``class_counts`` (5, C) int32 with
    planes true positives, false positives, misses, duplicates, other-class, and the two counts
    that say how many rows of the inputs were valid."""

    def __init__(self, det_match, det_iou, class_counts, det_count, gt_count):
        self.det_match, self.det_iou, self.class_counts = det_match, det_iou, class_counts
        self.det_count, self.gt_count = det_count, gt_count

    def summary(self, class_counts=None):
        """Host-side view (one device read): per class tp / fp / fn / duplicates / other_class
        and precision = tp / (tp + fp), recall = tp / (tp + fn) (NaN where undefined)."""
        c = (self.class_counts if class_counts is None else class_counts).cpu().numpy().astype(np.int64)
        with np.errstate(invalid="ignore", divide="ignore"):
            precision = c[0] / (c[0] + c[1])
            recall = c[0] / (c[0] + c[2])
        return {"tp": c[0], "fp": c[1], "fn": c[2], "duplicates": c[3], "other_class": c[4],
                "precision": precision, "recall": recall}


def comp_nms_gt(nms_boxes, gt_boxes, iou_thr=None, num_classes=None):
    """The validation comparison the reference describes at train.py:365-366 ("check iou of each
    nms box with each ground truth from val set, if above threshold compare classification") and
    leaves empty at train.py:279-280, as one pass of the matching kernels on the device.

    nms_boxes: the detections -- the 6-tuple of ``DetectNet.detect_device``, or ``(rows, count)``
    with rows (cap, 8) [image, x1, y1, x2, y2, objectness, class score, class id] as
    ``write_results`` / ``YoloV3Post.run`` emit them, or such a rows tensor alone (all valid).
    gt_boxes: a batch of true_box_grid (B, H, W, A, 5), or the 4-tuple of
    ``create_dataset.decode_true_box_grids_device``.  Both sides must be in the same pixel frame
    (the reference mode scales to params.im_height x im_width on both).
    iou_thr defaults to params.detection_iou.  Returns a ``MatchResult``; no host sync."""
    if isinstance(nms_boxes, (tuple, list)) and len(nms_boxes) >= 5:
        rows, count = detection_rows(nms_boxes)
    elif isinstance(nms_boxes, (tuple, list)):
        if len(nms_boxes) != 2:
            raise TypeError("nms_boxes: the outputs of DetectNet.detect_device, (rows, count), or a rows tensor")
        rows, count = nms_boxes
    else:
        rows, count = nms_boxes, None
    rows = _lib.as_f32(_lib.require_device(rows, "detection rows"), "detection rows")
    if rows.dim() != 2 or rows.shape[1] != 8:
        raise ValueError("detection rows must be (D, 8)")
    dev = rows.device
    if count is None:
        count = torch.tensor([rows.shape[0]], dtype=torch.int32, device=dev)
    count = _lib.require_device(count, "count").reshape(-1)[:1].to(device=dev, dtype=torch.int32)
    if isinstance(gt_boxes, (tuple, list)):
        g_index, g_boxes, g_class, g_count = gt_boxes
    else:
        g_index, g_boxes, g_class, g_count = create_dataset.decode_true_box_grids_device(gt_boxes)
    C = int(params.num_classes if num_classes is None else num_classes)
    thr = float(params.detection_iou if iou_thr is None else iou_thr)
    det_cap, gt_cap = max(int(rows.shape[0]), 1), int(g_boxes.shape[0])
    if rows.shape[0] == 0:                       # an empty list still reports the misses
        rows = torch.zeros((1, 8), dtype=torch.float32, device=dev)
        count = torch.zeros((1,), dtype=torch.int32, device=dev)
    det_match = torch.empty((det_cap,), dtype=torch.int32, device=dev)
    det_iou = torch.empty((det_cap,), dtype=torch.float32, device=dev)
    counts = torch.empty((_lib.YP_MATCH_PLANES, C), dtype=torch.int32, device=dev)
    L = _lib.load()
    ws = _lib.workspace(L.yp_match_detections_workspace_bytes(gt_cap), dev)
    with _lib.on_device(dev):
        rc = L.yp_match_detections(
            rows.data_ptr(), count.data_ptr(), det_cap, g_boxes.data_ptr(), g_class.data_ptr(),
            g_index.data_ptr(), g_count.data_ptr(), gt_cap, C, thr, det_match.data_ptr(),
            det_iou.data_ptr(), counts.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr(dev))
    _lib.check(rc, "yp_match_detections")
    return MatchResult(det_match, det_iou, counts, count, g_count)


def get_metrics(pred_boxes, pred_scores, pred_classes, pred_grid_indexes, true_grid, verbose=True):
    """train.py:331-340: the (image, row, col, anchor) locations of the true boxes next to the
    kept prediction indexes.  Returns the location array it prints."""
    uniques = create_dataset.true_box_locations(true_grid)
    if verbose:
        print('box locations in true grid', uniques)
        print('box locations in predictions', pred_grid_indexes)
    return uniques


class ValidationResult(list):
    """run_validation's list of per-batch tuples, plus ``metrics``: the per-class statistics of
    ``comp_nms_gt`` summed over every batch of the validation set (``MatchResult.summary``
    form), and ``class_counts`` (5, C) int64, the sums themselves."""
    metrics = None
    class_counts = None


def run_validation(val_dataset, model, verbose=True, iou_thr=None):
    """train.py:343-353: model -> predict_transform -> filter_boxes -> get_metrics per batch.
    Returns the list of (pred_boxes, pred_scores, pred_classes, pred_grid_indexes, locations)
    for the batches that produced any box.  Every batch (also one without boxes, whose ground
    truth then counts as missed) is also matched against its ground truth on the device
    (``comp_nms_gt``, the statistics train.py:365-366 asks for); the totals are in the returned
    list's ``metrics``."""
    if verbose:
        print('running validation set')
    out = ValidationResult()
    total = None
    if hasattr(model, "eval"):
        model.eval()
    with torch.no_grad():
        for val_x, val_grid in val_dataset:
            logits = model(val_x)
            dets = DetectNet.detect_device(logits)
            match = comp_nms_gt(dets, val_grid, iou_thr)
            c = match.class_counts.to(torch.int64)
            total = c if total is None else total + c
            pred_boxes, pred_scores, pred_classes, pred_grid_indexes = _unpad(dets)
            if pred_boxes.size == 0:
                continue
            loc = get_metrics(pred_boxes, pred_scores, pred_classes, pred_grid_indexes, val_grid,
                              verbose=verbose)
            out.append((pred_boxes, pred_scores, pred_classes, pred_grid_indexes, loc))
    if total is not None:
        out.class_counts = total.cpu().numpy()
        out.metrics = MatchResult(None, None, total, None, None).summary()
        if verbose:
            print('per class tp', out.metrics["tp"], 'fp', out.metrics["fp"], 'fn', out.metrics["fn"])
            print('precision', out.metrics["precision"], 'recall', out.metrics["recall"])
    return out
