"""The training-side callers of the detection path with the reference's names
(train.py:110-114 create_mask, :123-197 loss_custom, :200-270 loss_keras, :273-276 grad,
:331-340 get_metrics, :343-353 run_validation).

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
from .detect_net import DetectNet, _anchor_array, _device_tensor


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


def get_metrics(pred_boxes, pred_scores, pred_classes, pred_grid_indexes, true_grid, verbose=True):
    """train.py:331-340: the (image, row, col, anchor) locations of the true boxes next to the
    kept prediction indexes.  Returns the location array it prints."""
    uniques = create_dataset.true_box_locations(true_grid)
    if verbose:
        print('box locations in true grid', uniques)
        print('box locations in predictions', pred_grid_indexes)
    return uniques


def run_validation(val_dataset, model, verbose=True):
    """train.py:343-353: model -> predict_transform -> filter_boxes -> get_metrics per batch.
    Returns the list of (pred_boxes, pred_scores, pred_classes, pred_grid_indexes, locations)
    for the batches that produced any box."""
    if verbose:
        print('running validation set')
    out = []
    if hasattr(model, "eval"):
        model.eval()
    with torch.no_grad():
        for val_x, val_grid in val_dataset:
            logits = model(val_x)
            pred_boxes, pred_scores, pred_classes, pred_grid_indexes = DetectNet.detect(logits)
            if pred_boxes.size == 0:
                continue
            loc = get_metrics(pred_boxes, pred_scores, pred_classes, pred_grid_indexes, val_grid,
                              verbose=verbose)
            out.append((pred_boxes, pred_scores, pred_classes, pred_grid_indexes, loc))
    return out
