"""The decode as the epilogue of the kernel that produces the head (detect_net.py:67-68 followed
by :152-183 in one launch): bit for bit what predict_transform gives on the stored head, and
the head itself equals a float64 dense layer to rounding."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_decode_fused_behind_the_dense_layer_equals_the_two_step_path(pkg, dev):
    DN, P = pkg.DetectNet, pkg.params
    n_out = P.grid_height * P.grid_width * P.num_anchors * P.pred_vec_len          # 8450
    g = torch.Generator().manual_seed(5)
    for B, F in ((3, 64), (1, 7), (16, 32)):
        x = torch.randn((B, F), generator=g)
        w = torch.randn((F, n_out), generator=g) * 0.3
        b = torch.randn((n_out,), generator=g) * 0.5
        fused = DN.dense_relu_predict_transform(x.to(dev), w.to(dev), b.to(dev), return_head=True)
        head = fused[4]
        # the head: relu of the dense layer (float64 on the host as the yardstick)
        want_head = torch.relu(x.double() @ w.double() + b.double())
        assert torch.allclose(head.cpu().double(), want_head, rtol=1e-5, atol=1e-5)
        assert float(head.min()) >= 0.0 and float((head == 0).float().mean()) > 0.2     # relu did clip
        # the decode: identical bits to predict_transform on that head
        two_step = DN.predict_transform(head)
        for got, want in zip(fused[:4], two_step):
            assert got.shape == want.shape
            assert torch.equal(got, want)
        # and the fused call without the head output gives the same four tensors
        for got, want in zip(DN.dense_relu_predict_transform(x.to(dev), w.to(dev), b.to(dev)), two_step):
            assert torch.equal(got, want)


def test_fused_entry_rejects_what_it_does_not_support(pkg, dev, monkeypatch):
    DN, P = pkg.DetectNet, pkg.params
    n_out = P.grid_height * P.grid_width * P.num_anchors * P.pred_vec_len
    x = torch.zeros((2, 4), device=dev)
    with pytest.raises(ValueError):
        DN.dense_relu_predict_transform(x, torch.zeros((4, n_out - 1), device=dev), torch.zeros(n_out, device=dev))
    monkeypatch.setattr(P, "num_classes", 7)
    monkeypatch.setattr(P, "pred_vec_len", 12)
    n7 = P.grid_height * P.grid_width * P.num_anchors * 12
    with pytest.raises(pkg._lib.YoloPostError):
        DN.dense_relu_predict_transform(x, torch.zeros((4, n7), device=dev), torch.zeros(n7, device=dev))
