"""Backward of the reference decode and the fused loss_custom kernel (train.py:123-197,
detect_net.py:152-183) against a plain PyTorch restatement differentiated by autograd.

Floating-point bar: |got - want| <= 1e-5 * |want| + 1e-6 * (largest |want| of the tensor);
the second term covers entries that are differences of nearly equal products (softmax
backward), whose float32 reference itself carries that much rounding error against float64.

PARITY UNPINNED: no reference fixture exists for these functions and TensorFlow is absent;
the torch restatement is tied to the C oracle's forward decode in the first test below.
"""
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import np_oracle as npo
from oracle import torch_ref

HERE = os.path.dirname(os.path.abspath(__file__))
ANCHORS = np.array(((0.57273, 0.677385), (1.87446, 2.06253), (3.33843, 5.47434),
                    (7.88282, 3.52778), (9.77052, 9.16828)))
H = W = 13
A = C = 5
RTOL, ATOL_SCALE = 1e-5, 1e-6


def close(got, want, rtol=RTOL, scale=ATOL_SCALE):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    tol = rtol * np.abs(want) + scale * np.abs(want).max()
    bad = np.abs(got - want) > tol
    assert not bad.any(), "max excess %g at %d entries" % (
        (np.abs(got - want) - tol).max(), int(bad.sum()))


def make_case(B, seed):
    rs = np.random.RandomState(seed)
    pred = rs.normal(0, 1.5, (B, H * W * A * (5 + C))).astype(np.float32)
    grids = []
    for b in range(B):
        n = int(rs.randint(0, 12))
        y = np.stack([rs.uniform(0, 1920, n), rs.uniform(0, 1280, n), rs.uniform(8, 1200, n),
                      rs.uniform(8, 900, n), rs.randint(0, 5, n).astype(np.float64)], 1)
        grids.append(npo.create_true_box_grid(y, ANCHORS)[0])
    return pred, np.stack(grids)


def ref_loss_and_grad(pred, grid, dtype):
    p = torch.tensor(pred, dtype=dtype, requires_grad=True)
    out = torch_ref.loss_custom(p, torch.tensor(grid, dtype=dtype), H, W, A, C, ANCHORS)
    out[0].backward()
    return [float(o.detach()) for o in out], p.grad.numpy()


# ------------------------------------------------------------------------------ CPU

def test_torch_restatement_matches_c_oracle_forward():
    pred, _ = make_case(3, 1)
    want = oracle.r_predict_transform(pred, H, W, A, C, ANCHORS)
    got = torch_ref.predict_transform(torch.tensor(pred), H, W, A, C, ANCHORS)
    for g, w in zip(got, want):
        np.testing.assert_allclose(g.numpy(), w, rtol=1e-5, atol=1e-7)


def test_loss_known_answers():
    pred, grid = make_case(2, 2)
    _, _, obj, _ = oracle.r_predict_transform(pred, H, W, A, C, ANCHORS)
    # empty targets: only the no-object confidence term remains, 0.5 * sum(obj^2)
    total, conf, cls, coord = [float(v) for v in torch_ref.loss_custom(
        torch.tensor(pred), torch.zeros(2, H, W, A, 5), H, W, A, C, ANCHORS)]
    assert cls == 0 and coord == 0
    assert abs(total - 0.5 * float((obj.astype(np.float64) ** 2).sum())) < 1e-4 * total
    # float32 and float64 evaluations of a populated case agree
    l32, g32 = ref_loss_and_grad(pred, grid, torch.float32)
    l64, g64 = ref_loss_and_grad(pred, grid, torch.float64)
    assert abs(l32[0] - l64[0]) < 1e-5 * abs(l64[0])
    close(g32, g64)


# ------------------------------------------------------------------------------ GPU

@pytest.mark.gpu
def test_predict_transform_backward_matches_autograd(pkg, dev):
    pred, _ = make_case(4, 3)
    rs = np.random.RandomState(4)
    ups = [rs.normal(0, 1, (4, H, W, A, k)).astype(np.float32) for k in (2, 2, 1, C)]
    p_ref = torch.tensor(pred, requires_grad=True)
    outs = torch_ref.predict_transform(p_ref, H, W, A, C, ANCHORS)
    torch.autograd.backward(outs, [torch.tensor(u) for u in ups])
    p_gpu = torch.tensor(pred, device=dev, requires_grad=True)
    got = pkg.DetectNet.predict_transform(p_gpu)
    assert all(o.requires_grad for o in got)
    torch.autograd.backward(got, [torch.tensor(u, device=dev) for u in ups])
    assert p_gpu.grad.shape == p_gpu.shape
    close(p_gpu.grad.cpu().numpy(), p_ref.grad.numpy())
    # float64 autograd as the arbiter: the kernel is as close to it as the float32 reference is
    p64 = torch.tensor(pred, dtype=torch.float64, requires_grad=True)
    torch.autograd.backward(torch_ref.predict_transform(p64, H, W, A, C, ANCHORS),
                            [torch.tensor(u, dtype=torch.float64) for u in ups])
    close(p_gpu.grad.cpu().numpy(), p64.grad.numpy())


@pytest.mark.gpu
def test_predict_transform_backward_partial_outputs(pkg, dev):
    """A loss that uses only some outputs leaves the other upstream gradients absent."""
    pred, _ = make_case(2, 5)
    p_ref = torch.tensor(pred, requires_grad=True)
    c, w, o, s = torch_ref.predict_transform(p_ref, H, W, A, C, ANCHORS)
    ((o ** 2).sum() + (w * c).sum()).backward()
    p_gpu = torch.tensor(pred, device=dev, requires_grad=True)
    c, w, o, s = pkg.DetectNet.predict_transform(p_gpu)
    ((o ** 2).sum() + (w * c).sum()).backward()
    close(p_gpu.grad.cpu().numpy(), p_ref.grad.numpy())
    g = p_gpu.grad.cpu().numpy().reshape(-1, 5 + C)
    assert not g[:, 5:].any()                      # class logits got no gradient
    # no graph is recorded without requires_grad
    outs = pkg.DetectNet.predict_transform(torch.tensor(pred, device=dev))
    assert not any(t.requires_grad for t in outs)


@pytest.mark.gpu
def test_fused_loss_value_and_gradient(pkg, dev):
    pred, grid = make_case(6, 6)
    want_l, want_g = ref_loss_and_grad(pred, grid, torch.float32)
    l64, g64 = ref_loss_and_grad(pred, grid, torch.float64)
    p = torch.tensor(pred, device=dev, requires_grad=True)
    total, parts = pkg.train.loss_from_logits(p, torch.tensor(grid, device=dev), return_parts=True)
    total.backward()
    got_l = [float(total.detach())] + [float(v) for v in parts]
    for g, w, w64 in zip(got_l, want_l, l64):
        assert abs(g - w) <= 1e-5 * abs(w) + 1e-6 and abs(g - w64) <= 1e-5 * abs(w64) + 1e-6
    assert abs(got_l[0] - 0.5 * sum(got_l[1:])) <= 1e-6 * got_l[0]
    close(p.grad.cpu().numpy(), want_g)
    close(p.grad.cpu().numpy(), g64)
    # value only (no gradient requested), and run-to-run reproducibility of the reduction
    again = pkg.train.loss_keras(torch.tensor(grid, device=dev), torch.tensor(pred, device=dev))
    assert float(again) == got_l[0] and not again.requires_grad


@pytest.mark.gpu
def test_fused_loss_equals_composed_loss_on_gpu(pkg, dev):
    """The same loss written with torch ops on top of the differentiable predict_transform."""
    pred, grid = make_case(3, 7)
    tg = torch.tensor(grid, device=dev)
    p1 = torch.tensor(pred, device=dev, requires_grad=True)
    pkg.train.loss_from_logits(p1, tg).backward()
    p2 = torch.tensor(pred, device=dev, requires_grad=True)
    center, wh, obj, cls = pkg.DetectNet.predict_transform(p2)
    m = pkg.train.create_mask(tg)[..., :1]
    inter_wh = torch.clamp(torch.minimum(center + wh / 2, tg[..., 0:2] + tg[..., 2:4] / 2)
                           - torch.maximum(center - wh / 2, tg[..., 0:2] - tg[..., 2:4] / 2), min=0)
    inter = inter_wh[..., 0] * inter_wh[..., 1]
    iou = (inter / (wh[..., 0] * wh[..., 1] + tg[..., 2] * tg[..., 3] - inter)).unsqueeze(-1)
    det = (iou > 0.6).float()
    onehot = torch.nn.functional.one_hot(tg[..., 4].long(), C).float()
    loss = 0.5 * (((1 - det) * (1 - m) * obj ** 2 + 5 * m * (1 - obj) ** 2).sum()
                  + (m * (onehot - cls) ** 2).sum()
                  + (m * (tg[..., :4] - torch.cat([center, wh], -1)) ** 2).sum())
    loss.backward()
    close(p1.grad.cpu().numpy(), p2.grad.cpu().numpy())


@pytest.mark.gpu
def test_grad_through_a_model_and_weights(pkg, dev):
    """train.grad (train.py:273-276) with a small dense head, non-default loss weights."""
    torch.manual_seed(0)
    B, F = 5, 24
    model = torch.nn.Linear(F, H * W * A * (5 + C))
    x = torch.randn(B, F)
    _, grid = make_case(B, 8)
    old = (pkg.params.coord_loss_weight, pkg.params.obj_loss_weight, pkg.params.noobj_loss_weight)
    pkg.params.coord_loss_weight, pkg.params.obj_loss_weight, pkg.params.noobj_loss_weight = 2.0, 3.0, 0.5
    try:
        ref_loss = torch_ref.loss_custom(model(x), torch.tensor(grid), H, W, A, C, ANCHORS,
                                         coord_w=2.0, obj_w=3.0, noobj_w=0.5)[0]
        ref_grads = torch.autograd.grad(ref_loss, list(model.parameters()))
        gpu_model = torch.nn.Linear(F, H * W * A * (5 + C)).to(dev)
        gpu_model.load_state_dict(model.state_dict())
        loss, grads = pkg.train.grad(gpu_model, x.to(dev), torch.tensor(grid, device=dev))
    finally:
        pkg.params.coord_loss_weight, pkg.params.obj_loss_weight, pkg.params.noobj_loss_weight = old
    assert abs(float(loss) - float(ref_loss)) <= 2e-5 * abs(float(ref_loss))
    for g, w in zip(grads, ref_grads):
        close(g.cpu().numpy(), w.numpy(), rtol=1e-4, scale=1e-5)   # a (B x F) GEMM sits in between


@pytest.mark.gpu
def test_run_validation_and_get_metrics(pkg, dev):
    pred, grid = make_case(4, 9)
    pred[:, 4::10] += 6.0                       # push objectness up so that boxes survive
    dataset = [(torch.tensor(pred[:2], device=dev), torch.tensor(grid[:2], device=dev)),
               (torch.tensor(pred[2:], device=dev), torch.tensor(grid[2:], device=dev))]
    out = pkg.train.run_validation(dataset, lambda x: x, verbose=False)
    assert len(out) == 2
    for k, (boxes, scores, classes, index, loc) in enumerate(out):
        assert np.array_equal(loc, npo.true_box_locations(grid[2 * k:2 * k + 2]))
        want = oracle.r_detect(pred[2 * k:2 * k + 2], H, W, A, C, ANCHORS, 0.6, 0.5, 20, 1280, 1920)
        for b in range(2):
            assert np.array_equal(np.asarray(index[b]), want[3][b])
            assert np.array_equal(np.asarray(boxes[b]), want[0][b])
