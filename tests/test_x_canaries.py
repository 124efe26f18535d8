"""Out-of-bounds canaries: every buffer the library writes -- output rows, counts, statistics
and the workspace -- is an interior view of a larger allocation filled with a sentinel byte, and
after the run every byte outside the views must still hold it.  Calls go through the C ABI."""
import ctypes

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu

SENTINEL = 0xA5
PAD = 1 << 16


class Guarded:
    """nbytes of device memory, 256-byte aligned, with PAD sentinel bytes on either side."""

    def __init__(self, nbytes, dev):
        self.nbytes = int(nbytes)
        self.buf = torch.full((PAD + self.nbytes + PAD + 256,), SENTINEL, dtype=torch.uint8, device=dev)
        self.off = PAD + (-(self.buf.data_ptr() + PAD)) % 256
        self.ptr = self.buf.data_ptr() + self.off
        assert self.ptr % 256 == 0

    def view(self, dtype, shape):
        return self.buf[self.off:self.off + self.nbytes].view(dtype).view(shape)

    def intact(self):
        before, after = self.buf[:self.off], self.buf[self.off + self.nbytes:]
        return bool((before == SENTINEL).all().item()) and bool((after == SENTINEL).all().item())


def _descs(pkg, heads, inp_dim, C):
    L, util, syn = pkg._lib, pkg.util, pkg.synthetic
    return [util._head_desc(h, inp_dim, a, C) for h, a in zip(heads, syn.YOLOV3_ANCHORS)]


def _run_v3(pkg, dev, sets, inp_dim, conf, cap, dense):
    """yp_detect_yolov3_multi on guarded buffers; returns (per-set rows, counts, guards)."""
    L, lib, util, syn = pkg._lib, pkg._lib.load(), pkg.util, pkg.synthetic
    C, n_sets, B = 5, len(sets), sets[0][0].shape[0]
    flat = [d for heads in sets for d in _descs(pkg, heads, inp_dim, C)]
    descs = (L.YpHead * len(flat))(*flat)
    p = util._v3_params(B, C, conf, 0.4, batch_offset=2, dense=dense)
    need = lib.yp_detect_yolov3_workspace_bytes(descs, 3, n_sets * B, C)
    g_out = Guarded(n_sets * cap * 8 * 4, dev)
    g_cnt = Guarded(n_sets * 4, dev)
    g_st = Guarded(n_sets * 4 * B * C * 4, dev)
    g_ws = Guarded(need, dev)
    rc = lib.yp_detect_yolov3_multi(descs, 3, n_sets, ctypes.byref(p), g_out.ptr, cap, g_cnt.ptr, g_st.ptr,
                                    g_ws.ptr, need, L.stream_ptr(dev))
    assert rc == 0, lib.yp_strerror(rc)
    torch.cuda.synchronize()
    counts = g_cnt.view(torch.int32, (n_sets,)).cpu().numpy()
    rows = g_out.view(torch.float32, (n_sets, cap, 8)).cpu().numpy()
    return rows, counts, (g_out, g_cnt, g_st, g_ws)


@pytest.mark.parametrize("name,n_sets,B,inp_dim,conf,dense,cap_frac,clustered", [
    ("cfg2", 1, 64, 416, 0.5, False, 1.0, False),
    ("cfg2 x4 sets", 4, 16, 416, 0.5, False, 1.0, True),
    ("cfg5 slice, one-workgroup path", 1, 1, (1280, 1920), 0.001, False, 1.0, False),
    ("cfg5 slice, graph path", 1, 1, (1280, 1920), 0.001, True, 1.0, False),
    ("capacity overflow", 2, 4, 416, 0.3, False, 0.4, True),
    ("capacity overflow, graph path", 1, 2, 416, 0.001, True, 0.25, False),
])
def test_v3_writes_stay_inside_their_buffers(pkg, dev, name, n_sets, B, inp_dim, conf, dense, cap_frac, clustered):
    syn = pkg.synthetic
    cpu_sets, sets = [], []
    for k in range(n_sets):
        chunks = [syn.make_heads(min(16, B - lo), inp_dim, 5, seed=7000 + 100 * k + lo, clustered=clustered)
                  for lo in range(0, B, 16)]
        hs = [torch.cat([c[h] for c in chunks], 0).contiguous() for h in range(3)]
        cpu_sets.append(hs)
        sets.append([h.to(dev) for h in hs])
    wants = [oracle.n_detect([h.numpy() for h in hs], inp_dim, syn.YOLOV3_ANCHORS, 5, conf, 0.4)[0]
             for hs in cpu_sets]
    full = max(len(w) for w in wants)
    cap = max(2, int(full * cap_frac) & ~1)
    rows, counts, guards = _run_v3(pkg, dev, sets, inp_dim, conf, cap, dense)
    for g, what in zip(guards, ("output rows", "counts", "statistics", "workspace")):
        assert g.intact(), "%s: bytes outside the %s were overwritten" % (name, what)
    for k, want in enumerate(wants):
        want = want.copy()
        want[:, 0] += 2
        assert int(counts[k]) == len(want), name                       # the true total, also beyond cap
        n = min(cap, len(want))
        assert np.array_equal(rows[k, :n], want[:n]), name
    if cap_frac < 1.0:
        # rows past the capacity were not stored anywhere: the slab's tail still holds the sentinel
        tail = guards[0].view(torch.uint8, (n_sets, cap * 32))
        assert all(int(counts[k]) > cap for k in range(n_sets))
        assert guards[0].intact()


def test_reference_mode_writes_stay_inside_their_buffers(pkg, dev):
    L, lib, P = pkg._lib, pkg._lib.load(), pkg.params
    B, H, W, A, C, mb = 16, 13, 13, 5, 5, 20
    pred = pkg.synthetic.make_reference_head(B, seed=123, relu=True, obj_mean=0.5)
    d_pred = pred.to(dev)
    head = L.make_head(d_pred, H, W, [tuple(a) for a in np.asarray(P.YOLO_ANCHORS).reshape(-1, 2)])
    p = L.YpParams()
    p.mode, p.batch, p.num_classes, p.max_out = L.YP_MODE_REFERENCE, B, C, mb
    p.conf_thr, p.nms_thr, p.im_h, p.im_w = P.object_conf, P.nms_conf, P.im_height, P.im_width
    need = lib.yp_detect_reference_workspace_bytes(B, H, W, A)
    g = {"boxes": Guarded(B * mb * 16, dev), "scores": Guarded(B * mb * 4, dev),
         "classes": Guarded(B * mb * 4, dev), "index": Guarded(B * mb * 4, dev),
         "counts": Guarded(B * 4, dev), "surv": Guarded(B * 4, dev), "ws": Guarded(need, dev)}
    rc = lib.yp_detect_reference(ctypes.byref(head), ctypes.byref(p), g["boxes"].ptr, g["scores"].ptr,
                                 g["classes"].ptr, g["index"].ptr, g["counts"].ptr, g["surv"].ptr,
                                 g["ws"].ptr, need, L.stream_ptr(dev))
    assert rc == 0, lib.yp_strerror(rc)
    torch.cuda.synchronize()
    for name, guard in g.items():
        assert guard.intact(), "bytes outside %s were overwritten" % name
    ob, os_, oc, oi, surv = oracle.r_detect(pred.numpy(), H, W, A, C, np.asarray(P.YOLO_ANCHORS), P.object_conf,
                                            P.nms_conf, mb, P.im_height, P.im_width)
    counts = g["counts"].view(torch.int32, (B,)).cpu().numpy()
    boxes = g["boxes"].view(torch.float32, (B, mb, 4)).cpu().numpy()
    assert np.array_equal(g["surv"].view(torch.int32, (B,)).cpu().numpy(), surv)
    for b in range(B):
        assert counts[b] == len(ob[b]) and np.array_equal(boxes[b, :counts[b]], ob[b])
