"""Seeded synthetic detection-head tensors for tests and the benchmark.

The generator always runs on the CPU (``torch.Generator().manual_seed``) so that the CPU
oracle and the GPU path see identical bits; callers upload the result.

Value distributions (all float32, i.i.d.): t_x,t_y ~ N(0,1); t_w,t_h ~ N(0,0.5^2) clamped
to [-4,4]; objectness logit ~ N(obj_mean, obj_std^2) (default N(-3, 2^2), about 6.7 % of
anchors exceed 0); class logits ~ N(0,1).

``clustered=True`` instead places ``n_objects`` object centres per image and raises the
objectness of the anchors whose cell lies near a centre, with box offsets/sizes pulled
toward the object, so that surviving boxes overlap heavily as real detections do.
"""
import torch

# Standard YOLOv3 pixel anchors, (w, h) per head, heads ordered stride 32, 16, 8.
# The reference repository only carries the five YOLOv2 anchors of its 13x13 grid
# (params.py:20-22); these are the published YOLOv3 defaults.
YOLOV3_ANCHORS = (
    ((116, 90), (156, 198), (373, 326)),
    ((30, 61), (62, 45), (59, 119)),
    ((10, 13), (16, 30), (33, 23)),
)
YOLOV3_STRIDES = (32, 16, 8)


def head_shapes(inp_dim, strides=YOLOV3_STRIDES):
    """[(gh, gw)] per head for an int (square) or (H, W) input size."""
    in_h, in_w = (inp_dim, inp_dim) if isinstance(inp_dim, int) else inp_dim
    return [(in_h // s, in_w // s) for s in strides]


def anchors_per_image(inp_dim, num_anchors=3, strides=YOLOV3_STRIDES):
    return sum(gh * gw * num_anchors for gh, gw in head_shapes(inp_dim, strides))


def make_heads(batch, inp_dim, num_classes=5, seed=0, anchors=YOLOV3_ANCHORS,
               strides=YOLOV3_STRIDES, obj_mean=-3.0, obj_std=2.0, clustered=False,
               n_objects=30):
    """NCHW heads [(B, A*(5+C), gh, gw)] on the CPU, float32, head order = ``strides``."""
    g = torch.Generator().manual_seed(seed)
    V = 5 + num_classes
    in_h, in_w = (inp_dim, inp_dim) if isinstance(inp_dim, int) else inp_dim
    centres = None
    if clustered:
        centres = torch.rand(batch, n_objects, 2, generator=g)            # (x, y) in [0,1)
        obj_cls = torch.randint(0, num_classes, (batch, n_objects), generator=g)
    heads = []
    for (gh, gw), anc in zip(head_shapes(inp_dim, strides), anchors):
        A = len(anc)
        t = torch.randn(batch, A, V, gh, gw, generator=g)
        t[:, :, 2:4] = (t[:, :, 2:4] * 0.5).clamp_(-4.0, 4.0)
        t[:, :, 4] = t[:, :, 4] * obj_std + obj_mean
        if clustered:
            ys = (torch.arange(gh).float() + 0.5) / gh
            xs = (torch.arange(gw).float() + 0.5) / gw
            for b in range(batch):
                dx = xs.view(1, 1, gw) - centres[b, :, 0].view(-1, 1, 1)
                dy = ys.view(1, gh, 1) - centres[b, :, 1].view(-1, 1, 1)
                near = (dx.abs() < 1.5 / gw + 0.02) & (dy.abs() < 1.5 / gh + 0.02)
                hit = near.any(0)                                         # (gh, gw)
                which = near.float().argmax(0)                            # first centre
                t[b, :, 4][:, hit] += 6.0
                t[b, :, 2:4][:, :, hit] *= 0.2
                cls = obj_cls[b][which]                                   # (gh, gw)
                boost = torch.zeros(num_classes, gh, gw)
                boost.scatter_(0, cls.unsqueeze(0), 4.0)
                t[b, :, 5:] += (boost * hit.float()).unsqueeze(0)
        heads.append(t.reshape(batch, A * V, gh, gw).contiguous())
    return heads


def make_reference_head(batch, grid_h=13, grid_w=13, num_anchors=5, num_classes=5, seed=0,
                        obj_mean=1.0, obj_std=2.0, cls_scale=3.0, relu=False):
    """Raw head of the reference network, (B, H*W*A*(5+C)), memory order
    (b, row, col, anchor, attr) (detect_net.py:153).  ``relu=True`` clamps at zero as the
    reference's Dense(..., activation='relu') output does (detect_net.py:68)."""
    g = torch.Generator().manual_seed(seed)
    V = 5 + num_classes
    t = torch.randn(batch, grid_h, grid_w, num_anchors, V, generator=g)
    t[..., 2:4] = (t[..., 2:4] * 0.5).clamp_(-4.0, 4.0)
    t[..., 4] = t[..., 4] * obj_std + obj_mean
    t[..., 5:] = t[..., 5:] * cls_scale
    if relu:
        t = t.clamp_(min=0.0)
    return t.reshape(batch, -1).contiguous()
