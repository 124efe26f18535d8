"""Ground-truth grid codec with the reference's call surface (create_dataset.py:47-115 and
:278-329; the location extraction is also train.py:332-338).

``create_true_box_grid(y)`` keeps the reference's signature and return structure for one
image (numpy in, numpy out); ``encode_true_box_grids`` is the batched form that leaves the
grids on the GPU where the loss / validation code consumes them.  The work is done by the HIP
kernels of libyolopost.so; there is no CPU path.  Configuration is read from ``params`` at
call time, as in the reference.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from . import params


def _anchors():
    a = np.ascontiguousarray(np.asarray(params.YOLO_ANCHORS, dtype=np.float32).reshape(-1, 2))
    if a.shape[0] != int(params.num_anchors):
        raise ValueError("params.YOLO_ANCHORS does not hold params.num_anchors anchors")
    return a


def _device():
    if not torch.cuda.is_available():
        raise _lib.YoloPostError("no ROCm device available; this library has no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def encode_true_box_grids(objects, device=None):
    """Batched create_true_box_grid (create_dataset.py:47-115).

    objects: a list with one (n_b, 5) array-like [cx, cy, w, h, class] per image (original
    pixels; n_b may be 0), or a tuple ``(rows, offsets)`` of device tensors -- rows (total, 5)
    float64 and offsets (B+1,) int32 -- for callers that already hold the labels on the GPU.
    Returns (true_box_grid (B, H, W, A, 5), box_mask (B, H, W, A, 2)) float32 device tensors.
    """
    H, W, A = int(params.grid_height), int(params.grid_width), int(params.num_anchors)
    V = int(params.true_vec_len)
    if V != 5:
        raise ValueError("params.true_vec_len must be 5")
    if isinstance(objects, tuple):
        rows, offsets = objects
        _lib.require_device(rows, "rows")
        _lib.require_device(offsets, "offsets")
        if rows.dtype != torch.float64 or offsets.dtype != torch.int32:
            raise TypeError("rows must be float64 and offsets int32")
        rows, offsets = rows.contiguous(), offsets.contiguous()
        dev = rows.device
        B = offsets.numel() - 1
    else:
        dev = torch.device(device) if device is not None else _device()
        per = [np.asarray(o, dtype=np.float64).reshape(-1, V) for o in objects]
        B = len(per)
        off = np.zeros(B + 1, np.int32)
        if B:
            off[1:] = np.cumsum([len(p) for p in per])
        flat = np.concatenate(per, 0) if B and off[-1] else np.zeros((0, V), np.float64)
        rows = torch.from_numpy(np.ascontiguousarray(flat)).to(dev)
        offsets = torch.from_numpy(off).to(dev)
    grid = torch.empty((B, H, W, A, V), dtype=torch.float32, device=dev)
    mask = torch.empty((B, H, W, A, 2), dtype=torch.float32, device=dev)
    if B == 0:
        return grid, mask
    anc = _anchors()
    with _lib.on_device(dev):
        rc = _lib.load().yp_encode_true_box_grid(
            rows.data_ptr(), offsets.data_ptr(), B, H, W, A,
            anc.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
            int(params.im_height), int(params.im_width), grid.data_ptr(), mask.data_ptr(),
            _lib.stream_ptr(dev))
    _lib.check(rc, "yp_encode_true_box_grid")
    return grid, mask


def create_true_box_grid(y):
    """One image's labels -> (true_box_grid (H, W, A, 5), box_mask (H, W, A, 2)) float32 numpy
    arrays, as create_dataset.py:47-115 returns them."""
    grid, mask = encode_true_box_grids([y])
    return grid[0].cpu().numpy(), mask[0].cpu().numpy()


def decode_true_box_grids_device(true_box_grid):
    """Occupied slots of a batch of grids, on the device and without a host sync:
    (index (cap, 4) int32 [image, row, col, anchor], boxes (cap, 4) float32 [cx, cy, w, h] in
    original pixels, classes (cap,) float32, count (1,) int32); rows beyond count are
    unspecified.  Order is ascending (image, row, col, anchor), the np.unique order of
    create_dataset.py:303 / train.py:336."""
    H, W, A = int(params.grid_height), int(params.grid_width), int(params.num_anchors)
    g = true_box_grid if isinstance(true_box_grid, torch.Tensor) else \
        torch.as_tensor(np.asarray(true_box_grid))
    if not g.is_cuda:
        g = g.to(_device())
    g = _lib.as_f32(g, "true_box_grid").reshape(-1, H, W, A, 5)
    B, dev = g.shape[0], g.device
    if B == 0:
        raise ValueError("empty batch")
    cap = B * H * W * A
    index = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    boxes = torch.empty((cap, 4), dtype=torch.float32, device=dev)
    classes = torch.empty((cap,), dtype=torch.float32, device=dev)
    count = torch.empty((1,), dtype=torch.int32, device=dev)
    L = _lib.load()
    ws = _lib.workspace(L.yp_true_box_grid_workspace_bytes(B, H, W, A), dev)
    anc = _anchors()
    with _lib.on_device(dev):
        rc = L.yp_decode_true_box_grid(
            g.data_ptr(), B, H, W, A, anc.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
            int(params.im_height), int(params.im_width), index.data_ptr(), boxes.data_ptr(),
            classes.data_ptr(), cap, count.data_ptr(), ws.data_ptr(), ws.numel(),
            _lib.stream_ptr(dev))
    _lib.check(rc, "yp_decode_true_box_grid")
    return index, boxes, classes, count


def gather_grid_boxes(true_box_grid):
    """The boxes a (H, W, A, 5) grid (or a batch of them) encodes, back in original pixels:
    (boxes (n, 4) float32 [cx, cy, w, h], locations (n, 3) or (n, 4) int64) as numpy arrays --
    the arithmetic of create_dataset.py:294-328 without the drawing."""
    single = np.ndim(true_box_grid) == 4 if not isinstance(true_box_grid, torch.Tensor) \
        else true_box_grid.dim() == 4
    index, boxes, _, count = decode_true_box_grids_device(true_box_grid)
    n = int(count.item())
    loc = index[:n].cpu().numpy().astype(np.int64)
    return boxes[:n].cpu().numpy(), (loc[:, 1:] if single else loc)


def true_box_locations(true_box_grid):
    """Sorted unique (image, row, col, anchor) of the non-zero entries of a batch of grids,
    (n, 4) int64 numpy -- `uniques` in get_metrics, train.py:332-338."""
    index, _, _, count = decode_true_box_grids_device(true_box_grid)
    return index[:int(count.item())].cpu().numpy().astype(np.int64)
