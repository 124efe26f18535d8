"""Differentiable CPU restatement of the reference decode and training loss in plain PyTorch
-- TEST INFRASTRUCTURE ONLY (the checker for the backward / loss kernels, never a product path).

``predict_transform`` follows detect_net.py:152-183 and ``loss_custom`` train.py:123-197 line
by line with torch ops in place of the TensorFlow ones; gradients come from torch.autograd,
as the reference gets them from tf.GradientTape (train.py:273-276).  dtype follows the input
(float32 for the parity bar, float64 to measure the float32 rounding error itself).

PARITY UNPINNED: the reference holds no fixture for these functions and TensorFlow is absent.
"""
import torch


def predict_transform(pred, H, W, A, C, anchors):
    t = pred.reshape(-1, H, W, A, 5 + C)                                       # :153
    k = torch.arange(H * W)
    idx = torch.stack([k % H, k // H], 1).reshape(1, H, W, 1, 2).to(t.dtype)   # :155-162
    dims = torch.tensor([H, W], dtype=t.dtype).reshape(1, 1, 1, 1, 2)          # :163
    center = (torch.sigmoid(t[..., 0:2]) + idx) / dims                         # :166-168
    obj = torch.sigmoid(t[..., 4:5])                                           # :171
    anc = torch.as_tensor(anchors, dtype=torch.float32).to(t.dtype).reshape(1, 1, 1, A, 2)
    wh = (torch.exp(t[..., 2:4]) * anc) / dims                                 # :177
    cls = torch.softmax(t[..., 5:], -1)                                        # :180
    return center, wh, obj, cls


def create_mask(true_box_grid):
    return (true_box_grid != 0).to(true_box_grid.dtype)                        # train.py:110-114


def loss_custom(pred, true_box_grid, H, W, A, C, anchors, coord_w=1.0, obj_w=5.0, noobj_w=1.0,
                iou_thr=0.6):
    """Returns (total, confidence_sum, classification_sum, coordinates_sum)."""
    center, wh, obj, cls = predict_transform(pred, H, W, A, C, anchors)
    tg = true_box_grid.reshape(-1, H, W, A, 5).to(pred.dtype)
    dm = create_mask(tg)                                                       # :129
    pred_mins, pred_maxes = center - wh / 2., center + wh / 2.                 # :132-136
    true_xy, true_wh = tg[..., 0:2], tg[..., 2:4]
    true_mins, true_maxes = true_xy - true_wh / 2., true_xy + true_wh / 2.     # :138-142
    inter_wh = torch.clamp(torch.minimum(pred_maxes, true_maxes)
                           - torch.maximum(pred_mins, true_mins), min=0.)      # :145-148
    inter = inter_wh[..., 0] * inter_wh[..., 1]
    union = wh[..., 0] * wh[..., 1] + true_wh[..., 0] * true_wh[..., 1] - inter
    best = (inter / union).unsqueeze(4)                                        # :157-162
    det = (best > iou_thr).to(pred.dtype)                                      # :167
    no_obj = noobj_w * (1 - det) * (1 - dm[..., :1]) * obj ** 2                # :170-171
    obj_conf = obj_w * dm[..., :1] * (1 - obj) ** 2                            # :174
    cid = tg[..., 4].to(torch.int32).long()                                    # :177
    onehot = torch.zeros_like(cls)
    ok = (cid >= 0) & (cid < C)
    onehot.scatter_(-1, cid.clamp(0, C - 1).unsqueeze(-1), ok.unsqueeze(-1).to(cls.dtype))
    class_loss = dm[..., :1] * (onehot - cls) ** 2                             # :179
    boxes = torch.cat([center, wh], -1)                                        # :182
    coord = coord_w * dm[..., :1] * (tg[..., :4] - boxes) ** 2                 # :188
    s_conf, s_cls, s_coord = (no_obj + obj_conf).sum(), class_loss.sum(), coord.sum()
    return 0.5 * (s_conf + s_cls + s_coord), s_conf, s_cls, s_coord            # :190-195
