"""YP_FLAG_RESCALE: kept boxes mapped from the letterboxed network input back to the original
frame and clamped to it, after NMS (the step that follows write_results in the YOLOv3 tutorial;
the reference scales its boxes to the camera frame at detect_net.py:115-116, :133)."""
import numpy as np
import pytest
import torch

import oracle
from oracle import np_oracle as npo


def test_rescale_oracle_known_answers():
    """A 1920x1280 frame letterboxed into 416x416: scale 416/1920, 69.33 px of padding above and
    below, none at the sides."""
    det = np.array([[0, 0.0, 69.333336, 416.0, 346.66666, .9, .8, 1],
                    [0, 104.0, 208.0, 312.0, 208.0, .9, .8, 1],
                    [1, -5.0, 0.0, 500.0, 500.0, .9, .8, 2]], np.float32)
    out = npo.n_rescale(det, 416, (1280, 1920))
    assert out.dtype == np.float32
    assert np.allclose(out[0, 1:5], [0, 0, 1920, 1280], atol=2e-3)
    assert np.allclose(out[1, 1:5], [480, 640, 1440, 640], atol=2e-3)
    assert np.array_equal(out[2, 1:5], np.float32([0, 0, 1920, 1280]))       # clamped to the frame
    assert np.array_equal(out[:, [0, 5, 6, 7]], det[:, [0, 5, 6, 7]])        # nothing else moves
    # a frame with the input's own shape is the identity (up to the clamp)
    inside = np.float32([[0, 10.5, 20.25, 470.0, 319.0, .9, .8, 1]])
    assert np.array_equal(npo.n_rescale(inside, (320, 480), (320, 480)), inside)
    # portrait frame: padding at the sides
    tall = npo.n_rescale(np.float32([[0, 104, 0, 312, 416, 1, 1, 0]]), 416, (832, 416))
    assert np.allclose(tall[0, 1:5], [0, 0, 416, 832], atol=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("inp_dim,frame,B,conf,seed", [
    (416, (1280, 1920), 3, 0.5, 21),
    (416, (1920, 1280), 2, 0.3, 22),          # portrait
    ((320, 480), (1280, 1920), 2, 0.4, 23),   # same aspect ratio: no padding
    (416, (375, 1242), 2, 0.5, 24),           # wide strip: most boxes clamp at the borders
    (96, (33, 57), 1, 0.2, 25),               # upscaling network output (frame smaller than input)
])
def test_rescaled_rows_match_oracle(pkg, dev, inp_dim, frame, B, conf, seed):
    syn, util = pkg.synthetic, pkg.util
    heads = syn.make_heads(B, inp_dim, 5, seed=seed)
    plain, surv, _ = oracle.n_detect([h.numpy() for h in heads], inp_dim, syn.YOLOV3_ANCHORS, 5, conf, 0.4)
    want = npo.n_rescale(plain, inp_dim, frame)
    assert len(want) > 0
    dheads = [h.to(dev) for h in heads]
    eng = util.YoloV3Post(B, inp_dim, 5, conf, 0.4, device=dev, frame=frame)
    got = eng.detect(dheads).cpu().numpy()
    assert got.shape == want.shape
    assert np.array_equal(got, want)
    assert (got[:, [1, 3]] >= 0).all() and (got[:, [1, 3]] <= frame[1]).all()
    assert (got[:, [2, 4]] >= 0).all() and (got[:, [2, 4]] <= frame[0]).all()
    # without the frame the same engine configuration returns the rows in input pixels
    base = util.YoloV3Post(B, inp_dim, 5, conf, 0.4, device=dev).detect(dheads).cpu().numpy()
    assert np.array_equal(base, plain)


@pytest.mark.gpu
def test_rescale_through_the_multi_set_entry(pkg, dev):
    syn, util = pkg.synthetic, pkg.util
    frame = (1280, 1920)
    sets = [syn.make_heads(2, 416, 5, seed=30 + i) for i in range(3)]
    eng = util.YoloV3PostMulti(3, 2, 416, 5, 0.5, 0.4, device=dev, frame=frame)
    got = eng.detect([[h.to(dev) for h in s] for s in sets])
    for i, s in enumerate(sets):
        plain, _, _ = oracle.n_detect([h.numpy() for h in s], 416, syn.YOLOV3_ANCHORS, 5, 0.5, 0.4)
        assert np.array_equal(got[i].cpu().numpy(), npo.n_rescale(plain, 416, frame)), i
