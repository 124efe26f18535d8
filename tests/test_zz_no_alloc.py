"""yp_detect_yolov3_noalloc -- the entry point that allocates nothing (its workspace is a
static device array inside libyolopost.so) -- against yp_detect_yolov3, through the C ABI.

Two calls in a row on one stream, the second with a smaller input: every output bit of each
must equal the existing entry point's (which the parity tests tie to the CPU oracle), so what
the first call left in the shared scratch cannot leak into the second."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _descs(L, util, syn, heads, inp_dim, C):
    return (L.YpHead * len(heads))(*[util._head_desc(h, inp_dim, a, C)
                                     for h, a in zip(heads, syn.YOLOV3_ANCHORS)])


def _run(pkg, dev, B, inp_dim, seed, conf, noalloc, clustered=False, dense=False):
    """One call through the C ABI; returns (rows[:count], stats) as numpy."""
    L, lib, util, syn = pkg._lib, pkg._lib.load(), pkg.util, pkg.synthetic
    C = 5
    heads = [h.to(dev) for h in syn.make_heads(B, inp_dim, C, seed=seed, clustered=clustered)]
    descs = _descs(L, util, syn, heads, inp_dim, C)
    p = util._v3_params(B, C, conf, 0.4, batch_offset=5, dense=dense)
    cap = B * syn.anchors_per_image(inp_dim)
    out = torch.full((cap, 8), -1.0, dtype=torch.float32, device=dev)
    count = torch.full((1,), -1, dtype=torch.int32, device=dev)
    stats = torch.full((4, B, C), -1, dtype=torch.int32, device=dev)
    stream = L.stream_ptr(dev)
    if noalloc:
        # torch's counter only sees torch's own allocator; the device's free bytes (hipMemGetInfo)
        # also move when the library itself allocates or frees
        torch.cuda.synchronize()
        before, free_before = torch.cuda.memory_allocated(dev), torch.cuda.mem_get_info(dev)[0]
        rc = lib.yp_detect_yolov3_noalloc(descs, 3, ctypes.byref(p), out.data_ptr(), cap,
                                          count.data_ptr(), stats.data_ptr(), stream)
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated(dev) == before
        assert torch.cuda.mem_get_info(dev)[0] == free_before
    else:
        need = lib.yp_detect_yolov3_workspace_bytes(descs, 3, B, C)
        ws = L.workspace(need, dev)
        rc = lib.yp_detect_yolov3(descs, 3, ctypes.byref(p), out.data_ptr(), cap, count.data_ptr(),
                                  stats.data_ptr(), ws.data_ptr(), ws.numel(), stream)
    assert rc == 0, lib.yp_strerror(rc)
    torch.cuda.synchronize()
    d = int(count.item())
    return out[:d].cpu().numpy(), stats.cpu().numpy()


def test_noalloc_entry_matches_existing_entry_on_two_calls_in_a_row(pkg, dev):
    L, lib = pkg._lib, pkg._lib.load()
    assert lib.yp_noalloc_scratch_bytes() == 16 << 20
    assert lib.yp_noalloc_prepare() == 0
    # first call: 8 images at 416x416, conf 0.001 (near-all survive: the large-segment arrays of
    # the scratch are written too); second call: 2 images at 96x96, conf 0.5 -- far smaller
    big = _run(pkg, dev, 8, 416, 71, 0.001, noalloc=True)
    small = _run(pkg, dev, 2, 96, 72, 0.5, noalloc=True, clustered=True)
    big_ref = _run(pkg, dev, 8, 416, 71, 0.001, noalloc=False)
    small_ref = _run(pkg, dev, 2, 96, 72, 0.5, noalloc=False, clustered=True)
    for got, want in ((big, big_ref), (small, small_ref)):
        assert got[0].shape == want[0].shape and got[0].shape[0] > 0
        assert np.array_equal(got[0], want[0])
        # survivors and kept rows per (image, class); the evaluated-pairs planes are not compared:
        # which pairs the large-segment path prunes depends on the order its atomics hand out
        assert np.array_equal(got[1][:2], want[1][:2])
    # and once more the other way round: small, then big
    assert np.array_equal(_run(pkg, dev, 2, 96, 72, 0.5, noalloc=True, clustered=True)[0], small_ref[0])
    assert np.array_equal(_run(pkg, dev, 8, 416, 71, 0.001, noalloc=True)[0], big_ref[0])


def test_noalloc_entry_with_the_dense_flag_on_calls_of_different_size(pkg, dev):
    """YP_FLAG_DENSE_NMS through the entry that allocates nothing: the graph arrays of the static
    scratch are used by a call with segments of ~2 000 candidates, then by a smaller one."""
    lib = pkg._lib.load()
    assert lib.yp_noalloc_prepare() == 0
    for B, inp_dim, seed in ((8, 416, 81), (3, 256, 82), (8, 416, 83)):
        got = _run(pkg, dev, B, inp_dim, seed, 0.001, noalloc=True, dense=True)
        want = _run(pkg, dev, B, inp_dim, seed, 0.001, noalloc=False, dense=False)
        assert got[1][0].max() > 768                     # segments large enough for the graph kernels
        assert got[0].shape == want[0].shape and np.array_equal(got[0], want[0])
        assert np.array_equal(got[1][:2], want[1][:2])


def test_noalloc_entry_rejects_problems_beyond_its_scratch(pkg, dev):
    L, lib, util, syn = pkg._lib, pkg._lib.load(), pkg.util, pkg.synthetic
    B, C = 64, 5                                   # 64 images at 416x416 need ~51 MB of workspace
    heads = [h.to(dev) for h in syn.make_heads(B, 416, C, seed=1)]
    descs = _descs(L, util, syn, heads, 416, C)
    assert lib.yp_detect_yolov3_workspace_bytes(descs, 3, B, C) > (16 << 20)
    p = util._v3_params(B, C, 0.5, 0.4)
    out = torch.full((16, 8), -1.0, dtype=torch.float32, device=dev)
    count = torch.full((1,), -7, dtype=torch.int32, device=dev)
    rc = lib.yp_detect_yolov3_noalloc(descs, 3, ctypes.byref(p), out.data_ptr(), 16, count.data_ptr(),
                                      None, L.stream_ptr(dev))
    assert rc == -4 and b"scratch" in lib.yp_strerror(rc)
    torch.cuda.synchronize()
    assert int(count.item()) == -7 and float(out.min()) == -1.0      # nothing was launched
