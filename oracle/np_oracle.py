"""Independent numpy restatement of the post-processing path -- TEST INFRASTRUCTURE ONLY.

Written separately from ``oracle/ypo.c`` (vectorised numpy float32 + plain Python loops
for the greedy NMS) so the two can cross-check each other.  numpy's exp is not the
bit-specified ``ypo_exp_spec``; decoded values therefore agree with the C oracle to a
few ulp (tests use 1e-5 relative), while the NMS routines, which use only + - * / and
comparisons, agree exactly when fed identical boxes.

Mode R follows /root/reference/detect_net.py:152-183 (decode) and :90-146 (filter/NMS);
Mode N follows the YOLOv3 tutorial semantics (not present in the reference); the ground-truth
grid codec at the bottom follows create_dataset.py:47-115 and :294-328.

PARITY UNPINNED: the reference holds no test, fixture or golden vector for any of these
functions and cannot be run here (TensorFlow is absent); the restatements are anchored by
hand-computed known answers and by each other.
"""
import numpy as np

f32 = np.float32


def sigmoid(x):
    x = np.asarray(x, f32)
    with np.errstate(over="ignore"):
        return (f32(1.0) / (f32(1.0) + np.exp(-x))).astype(f32)


# ------------------------------------------------------------------------------ Mode R

def r_predict_transform(pred, H, W, A, C, anchors):
    """detect_net.py:152-183."""
    V = 5 + C
    t = np.asarray(pred, f32).reshape(-1, H, W, A, V)                      # :153
    k = np.arange(H * W)
    idx = np.stack([k % H, k // H], 1).reshape(1, H, W, 1, 2).astype(f32)  # :155-162
    dims = np.array([H, W], f32).reshape(1, 1, 1, 1, 2)                    # :163
    center = (sigmoid(t[..., 0:2]) + idx) / dims                           # :166-168
    obj = sigmoid(t[..., 4:5])                                             # :171
    anc = np.asarray(anchors, f32).reshape(1, 1, 1, A, 2)                  # :173-174
    with np.errstate(over="ignore"):
        wh = (np.exp(t[..., 2:4]) * anc) / dims                            # :177
    z = t[..., 5:5 + C]
    e = np.exp(z - z.max(-1, keepdims=True))
    s = np.zeros(e.shape[:-1] + (1,), f32)
    for j in range(C):                                                     # sequential sum
        s = s + e[..., j:j + 1]
    cls = e / s                                                            # :180
    return center.astype(f32), wh.astype(f32), obj.astype(f32), cls.astype(f32)


def tf_iou(bi, bj):
    bi, bj = np.asarray(bi, f32), np.asarray(bj, f32)
    ymin_i, ymax_i = min(bi[0], bi[2]), max(bi[0], bi[2])
    xmin_i, xmax_i = min(bi[1], bi[3]), max(bi[1], bi[3])
    ymin_j, ymax_j = min(bj[0], bj[2]), max(bj[0], bj[2])
    xmin_j, xmax_j = min(bj[1], bj[3]), max(bj[1], bj[3])
    area_i = f32(ymax_i - ymin_i) * f32(xmax_i - xmin_i)
    area_j = f32(ymax_j - ymin_j) * f32(xmax_j - xmin_j)
    if area_i <= 0 or area_j <= 0:
        return f32(0.0)
    dh = max(f32(min(ymax_i, ymax_j) - max(ymin_i, ymin_j)), f32(0.0))
    dw = max(f32(min(xmax_i, xmax_j) - max(xmin_i, xmin_j)), f32(0.0))
    inter = f32(dh * dw)
    return f32(inter / f32(f32(area_i + area_j) - inter))


def tf_nms(boxes, scores, max_out, thr):
    """Greedy NMS as tf.image.non_max_suppression (detect_net.py:136)."""
    boxes = np.asarray(boxes, f32).reshape(-1, 4)
    scores = np.asarray(scores, f32)
    order = sorted(range(len(scores)), key=lambda i: (-float(scores[i]), i))
    sel = []
    thr = f32(thr)
    for i in order:
        if len(sel) >= max_out:
            break
        if all(not (tf_iou(boxes[i], boxes[s]) > thr) for s in sel):
            sel.append(i)
    return np.array(sel, np.int32)


def r_filter_boxes(center, wh, obj, cls, conf_thr, nms_thr, max_boxes, im_h, im_w):
    """detect_net.py:90-146."""
    mins = center - wh / f32(2.0)                                          # :95
    maxes = center + wh / f32(2.0)                                         # :96
    boxes = np.concatenate([mins[..., 1:2], mins[..., 0:1],
                            maxes[..., 1:2], maxes[..., 0:1]], -1)         # :97-102
    sc = obj * cls                                                         # :107
    bcls = sc.argmax(-1)                                                   # :108
    bsc = sc.max(-1)                                                       # :109
    mask = bsc >= f32(conf_thr)                                            # :111
    dims = np.array([im_h, im_w, im_h, im_w], f32).reshape(1, 4)           # :115-116
    out = ([], [], [], [])
    surv = []
    for b in range(center.shape[0]):                                       # :125
        bb = boxes[b][mask[b]] * dims                                      # :126,133
        ss = bsc[b][mask[b]]
        cc = bcls[b][mask[b]].astype(np.int64)
        surv.append(len(ss))
        idx = tf_nms(bb, ss, max_boxes, nms_thr)                           # :136
        out[0].append(bb[idx]); out[1].append(ss[idx])                     # :137-139
        out[2].append(cc[idx]); out[3].append(idx)
    return out[0], out[1], out[2], out[3], np.array(surv, np.int32)


# ------------------------------------------------------------------------------ Mode N

def n_predict_transform(head, inp_dim, anchors_px, num_classes):
    head = np.asarray(head, f32)
    B, ch, gh, gw = head.shape
    A, C = len(anchors_px), num_classes
    V = 5 + C
    in_h, in_w = (inp_dim, inp_dim) if np.isscalar(inp_dim) else inp_dim
    sx, sy = in_w // gw, in_h // gh
    t = head.reshape(B, A, V, gh * gw).transpose(0, 3, 1, 2)               # (B, cell, A, V)
    cell = np.arange(gh * gw)
    col = (cell % gw).astype(f32).reshape(1, -1, 1)
    row = (cell // gw).astype(f32).reshape(1, -1, 1)
    anc = np.asarray(anchors_px, np.float64).reshape(A, 2)
    aw = (anc[:, 0] / sx).astype(f32).reshape(1, 1, A)
    ah = (anc[:, 1] / sy).astype(f32).reshape(1, 1, A)
    out = np.empty((B, gh * gw, A, V), f32)
    out[..., 0] = (sigmoid(t[..., 0]) + col) * f32(sx)
    out[..., 1] = (sigmoid(t[..., 1]) + row) * f32(sy)
    with np.errstate(over="ignore"):
        out[..., 2] = (np.exp(t[..., 2]) * aw) * f32(sx)
        out[..., 3] = (np.exp(t[..., 3]) * ah) * f32(sy)
    out[..., 4] = sigmoid(t[..., 4])
    out[..., 5:] = sigmoid(t[..., 5:])
    return out.reshape(B, gh * gw * A, V)


def n_iou(b1, b2):
    b1, b2 = np.asarray(b1, f32), np.asarray(b2, f32)
    one, zero = f32(1.0), f32(0.0)
    dw = f32(f32(min(b1[2], b2[2]) - max(b1[0], b2[0])) + one)
    dh = f32(f32(min(b1[3], b2[3]) - max(b1[1], b2[1])) + one)
    inter = f32(max(dw, zero) * max(dh, zero))
    a1 = f32(f32(f32(b1[2] - b1[0]) + one) * f32(f32(b1[3] - b1[1]) + one))
    a2 = f32(f32(f32(b2[2] - b2[0]) + one) * f32(f32(b2[3] - b2[1]) + one))
    with np.errstate(divide="ignore", invalid="ignore"):
        return f32(inter / f32(f32(a1 + a2) - inter))


def n_write_results(rows, confidence, num_classes, nms_conf=0.4):
    rows = np.asarray(rows, f32)
    B, N, V = rows.shape
    C = num_classes
    out = []
    for b in range(B):
        p = rows[b]
        obj = p[:, 4]
        keep = (obj > f32(confidence)) & (obj != 0)
        p = p[keep]
        if len(p) == 0:
            continue
        hw, hh = p[:, 2] / f32(2), p[:, 3] / f32(2)
        box = np.stack([p[:, 0] - hw, p[:, 1] - hh, p[:, 0] + hw, p[:, 1] + hh], 1)
        cscore = p[:, 5:5 + C].max(1)
        cid = p[:, 5:5 + C].argmax(1)
        for c in sorted(set(cid.tolist())):
            sel = np.nonzero(cid == c)[0]
            order = sorted(sel.tolist(), key=lambda i: (-float(p[i, 4]), i))
            alive = list(order)
            i = 0
            while i < len(alive):
                a = alive[i]
                alive = alive[:i + 1] + [j for j in alive[i + 1:]
                                         if n_iou(box[a], box[j]) < f32(nms_conf)]
                i += 1
            for a in alive:
                out.append([b, box[a, 0], box[a, 1], box[a, 2], box[a, 3],
                            p[a, 4], cscore[a], c])
    return np.array(out, f32).reshape(-1, 8)


# ------------------------------------------------------ ground-truth grid codec (reference)

def create_true_box_grid(y, anchors, grid_h=13, grid_w=13, im_h=1280, im_w=1920, true_vec_len=5):
    """create_dataset.py:47-115, restated line by line (numpy float64 with float32 anchors, as
    in the reference, where `y` is np.array(list of Python floats) and the anchors are the
    float32 tensor of get_anchors()).  y: (n, 5) [cx, cy, w, h, class] in original pixels.
    Objects whose cell lies outside the grid are skipped (the reference would raise an
    IndexError for an index past the end and wrap a negative one)."""
    A = len(anchors)
    true_box_grid = np.zeros((grid_h, grid_w, A, true_vec_len), np.float32)          # :49
    box_mask = np.zeros((grid_h, grid_w, A, 2), np.float32)                          # :50
    mask_vec = np.ones((2), np.float32)
    y = np.asarray(y, np.float64).reshape(-1, true_vec_len)
    if y.shape[0] == 0:                                                              # :52
        return true_box_grid, box_mask
    orig_size = np.array([im_w, im_h])                                               # :56
    center_boxes = y[:, :2] / orig_size                                              # :57
    boxes_wh = y[:, 2:4] / orig_size                                                 # :59
    conv_size = np.array([grid_w, grid_h])                                           # :61
    center_boxes = center_boxes * conv_size                                          # :63
    boxes_wh = boxes_wh * conv_size                                                  # :64
    true_boxes = np.concatenate([center_boxes, boxes_wh, y[:, 4:]], axis=1)          # :66
    anchors = np.asarray(anchors, np.float32)                                        # :67
    for obj in range(y.shape[0]):                                                    # :70
        best_iou = 0
        best_anchor = 0
        box = true_boxes[obj]
        i = np.floor(box[1]).astype('int')                                           # :74 row
        j = np.floor(box[0]).astype('int')                                           # :75 column
        for k, anchor in enumerate(anchors):                                         # :79
            box_max = box[2:4] / 2
            box_min = -box_max
            anchor_max = anchor / 2
            anchor_min = -anchor_max
            intersect_mins = np.maximum(box_min, anchor_min)
            intersect_maxes = np.minimum(box_max, anchor_max)
            intersect_wh = np.maximum(intersect_maxes - intersect_mins, 0)
            intersect_area = intersect_wh[0] * intersect_wh[1]
            box_area = box[2] * box[3]
            anchor_area = anchor[0] * anchor[1]
            with np.errstate(divide="ignore", invalid="ignore"):
                iou = intersect_area / (box_area + anchor_area - intersect_area)     # :94
            if iou > best_iou:
                best_iou = iou
                best_anchor = k
        if best_iou > 0 and 0 <= i < grid_h and 0 <= j < grid_w:                     # :100
            with np.errstate(divide="ignore", invalid="ignore"):
                adjusted_box = np.array(
                    [box[0] - j, box[1] - i,
                     np.log(box[2] / anchors[best_anchor][0]),
                     np.log(box[3] / anchors[best_anchor][1]),
                     box[4]], dtype=np.float32)                                      # :101-110
            true_box_grid[i, j, best_anchor] = adjusted_box                          # :111
            box_mask[i, j, best_anchor] = mask_vec                                   # :112
    return true_box_grid, box_mask


def true_box_locations(true_grid):
    """The index extraction shared by gather_grid_boxes (create_dataset.py:294-304) and
    get_metrics (train.py:332-338): unique leading indices of the non-zero entries, sorted.
    true_grid (..., 5): returns (n, ndim-1) int64."""
    g = np.asarray(true_grid)
    idx = np.argwhere(g != 0)[:, :g.ndim - 1]
    return np.unique(idx, axis=0) if len(idx) else np.zeros((0, g.ndim - 1), np.int64)


def decode_true_box_grid(true_grid, anchors, im_h=1280, im_w=1920, exp=None):
    """gather_grid_boxes, create_dataset.py:297-328: the boxes of one (H, W, A, 5) grid back in
    original pixels, (n, 4) float32 [cx, cy, w, h] ordered by (row, col, anchor), plus their
    locations.  `exp` lets the caller supply the bit-specified float32 exp."""
    g = np.asarray(true_grid, f32)
    H, W = g.shape[0], g.shape[1]
    uniq = true_box_locations(g)                                                     # :297-304
    if len(uniq) == 0:
        return np.zeros((0, 4), f32), uniq
    vec = g[uniq[:, 0], uniq[:, 1], uniq[:, 2]]                                      # :316
    offs = np.stack([uniq[:, 1], uniq[:, 0]], 1).astype(f32)                         # :306-314 (x=j, y=i)
    centre = vec[:, :2] + offs                                                       # :318
    anc = np.asarray(anchors, f32)[uniq[:, 2]]                                       # :320
    e = np.exp(vec[:, 2:4]) if exp is None else exp(vec[:, 2:4])                     # :321
    wh = (e * anc).astype(f32)                                                       # :322
    out = np.concatenate([centre, wh], 1).astype(f32)                                # :326
    conv = np.array([W, H, W, H], f32)
    orig = np.array([im_w, im_h, im_w, im_h], f32)
    return ((out / conv) * orig).astype(f32), uniq                                   # :327-328


def n_rescale(det, inp_dim, frame):
    """Detections (D, 8) in network-input pixels -> the original frame, as the YOLOv3 tutorial
    does after write_results (and as the reference scales its boxes to the camera frame,
    detect_net.py:115-116, :133): scale = min(in_w/im_w, in_h/im_h), letterbox pads
    (in - scale*im)/2, every corner (v - pad)/scale clamped to [0, im].  All in binary32, in
    this order."""
    f = np.float32
    in_h, in_w = (inp_dim, inp_dim) if np.isscalar(inp_dim) else inp_dim
    im_h, im_w = f(frame[0]), f(frame[1])
    in_h, in_w = f(in_h), f(in_w)
    sx, sy = in_w / im_w, in_h / im_h
    scale = sx if sx < sy else sy
    pad_x = (in_w - scale * im_w) / f(2.0)
    pad_y = (in_h - scale * im_h) / f(2.0)
    out = np.array(det, dtype=np.float32, copy=True)
    for col, pad, top in ((1, pad_x, im_w), (2, pad_y, im_h), (3, pad_x, im_w), (4, pad_y, im_h)):
        out[:, col] = np.minimum(np.maximum((out[:, col] - pad) / scale, f(0.0)), top)
    return out


# ------------------------------------------------------------- detections against ground truth

MATCH_NONE, MATCH_DUPLICATE, MATCH_OTHER_CLASS = -1, -2, -3


def match_iou(det, gt):
    """Plain geometric IoU (no +1 term) of a corner box det [x1, y1, x2, y2] with a centre/size
    box gt [cx, cy, w, h], binary32 in a fixed operation order."""
    d = [f32(v) for v in det]
    g = [f32(v) for v in gt]
    hw, hh = g[2] * f32(0.5), g[3] * f32(0.5)
    gx1, gy1, gx2, gy2 = g[0] - hw, g[1] - hh, g[0] + hw, g[1] + hh
    ix = max(min(d[2], gx2) - max(d[0], gx1), f32(0.0))
    iy = max(min(d[3], gy2) - max(d[1], gy1), f32(0.0))
    inter = f32(ix * iy)
    area_d = f32(f32(d[2] - d[0]) * f32(d[3] - d[1]))
    area_g = f32(f32(gx2 - gx1) * f32(gy2 - gy1))
    uni = f32(f32(area_d + area_g) - inter)
    return f32(inter / uni) if uni > 0 else f32(0.0)


def match_detections(rows, gt_index, gt_boxes, gt_class, num_classes, iou_thr):
    """The comparison train.py:365-366 describes and the empty comp_nms_gt (train.py:279-280)
    was meant to hold, by the PASCAL-VOC rule, as a sequential loop: detections in list order;
    the candidate of a detection is the ground-truth box of its image and class with the highest
    IoU (first among equals); true positive when that IoU >= iou_thr and the box is still free.
    Returns (det_match (D,) int32, det_iou (D,) float32, class_counts (5, C) int32 with planes
    true positives, false positives, misses, duplicates, other-class)."""
    rows = np.asarray(rows, f32).reshape(-1, 8)
    gt_index = np.asarray(gt_index, np.int64).reshape(-1, 4)
    gt_boxes = np.asarray(gt_boxes, f32).reshape(-1, 4)
    gt_class = np.asarray(gt_class, f32).reshape(-1)
    thr = f32(iou_thr)
    C = int(num_classes)
    taken = np.zeros(len(gt_boxes), bool)
    det_match = np.full(len(rows), MATCH_NONE, np.int32)
    det_iou = np.zeros(len(rows), f32)
    counts = np.zeros((5, C), np.int32)

    def bump(plane, cls):
        if 0 <= cls < C:
            counts[plane, int(cls)] += 1

    for d, r in enumerate(rows):
        img, cls = int(r[0]), r[7]
        best_any, best_same, best_other, cand = f32(0), f32(0), f32(0), -1
        for g in np.flatnonzero(gt_index[:, 0] == img):
            iou = match_iou(r[1:5], gt_boxes[g])
            best_any = max(best_any, iou)
            if gt_class[g] == cls:
                if iou > best_same:
                    best_same, cand = iou, int(g)
            elif iou > best_other:
                best_other = iou
        det_iou[d] = best_any
        if cand >= 0 and best_same >= thr:
            if not taken[cand]:
                taken[cand] = True
                det_match[d] = cand
                bump(0, cls)
                continue
            det_match[d] = MATCH_DUPLICATE
            bump(3, cls)
        elif best_other >= thr:
            det_match[d] = MATCH_OTHER_CLASS
            bump(4, cls)
        bump(1, cls)
    for g in np.flatnonzero(~taken):
        bump(2, gt_class[g])
    return det_match, det_iou, counts


def precision_recall(class_counts):
    """Per class (precision, recall) from the count planes; NaN where undefined."""
    c = np.asarray(class_counts, np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        return c[0] / (c[0] + c[1]), c[0] / (c[0] + c[2])
