// decode_grad.hip -- backward pass of the reference-mode anchor decode for MI355X (gfx950).
//
// loss_custom / loss_keras (train.py:123-197, :200-260) differentiate through
// DetectNet.predict_transform (detect_net.py:152-183).  Given the upstream gradients of its
// four outputs this kernel produces the gradient of the raw head:
//
//   center = (sigmoid(t01) + idx) / dims   ->  d t01 = (g / dims) * s * (1 - s)
//   wh     = exp(t23) * anchor / dims      ->  d t23 = g * wh
//   obj    = sigmoid(t4)                   ->  d t4  = g * o * (1 - o)
//   cls    = softmax(t5..)                 ->  d z_j = p_j * (g_j - sum_k g_k p_k)
//
// It is a pure streaming kernel (reads the head and the gradients once, writes the head
// gradient once); one thread owns one (cell, anchor) slot, so a wave covers one contiguous
// span of 64*(5+C) floats of every tensor it touches.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "yolopost.h"
#include "yp_math.h"
#include "yp_tile.h"

namespace {

struct GradAnchors {
    float anc[2 * YP_MAX_ANCHORS];
};

constexpr int kGradMaxClasses = 64;

// CS > 0: the class count is known at compile time and the softmax probabilities are held in
// registers (evaluated once); CS == 0 takes the count from the argument and re-evaluates them.
template <bool STAGED, int CS>
__global__ void __launch_bounds__(kTileRows)
k_decode_grid_ref_backward(const float *__restrict__ pred, int H, int W, int A, int C_arg, GradAnchors at,
                           const float *__restrict__ g_center, const float *__restrict__ g_wh,
                           const float *__restrict__ g_obj, const float *__restrict__ g_cls,
                           float *__restrict__ g_pred, long long total)
{
    // STAGED: [0, 256 V) the raw rows, later the gradient rows; [256 V, 256 (V + C)) the upstream
    // class gradients (yp_tile.h)
    extern __shared__ float s_tile[];
    const long long tile0 = (long long)blockIdx.x * kTileRows;
    const long long o = tile0 + threadIdx.x;
    const int C = CS ? CS : C_arg;
    const int V = 5 + C;
    float *s_gcls = s_tile + kTileRows * V;
    if (STAGED) {
        if (g_cls) yp_tile_load(g_cls + tile0 * C, (total - tile0) * C, C, s_gcls);
        yp_tile_load(pred + tile0 * V, (total - tile0) * V, V, s_tile);
    }
    float gt[STAGED ? kTileMaxWidth : 5];
    if (o < total) {
        const int a = (int)(o % A);
        const float *t = STAGED ? s_tile + threadIdx.x * V : pred + o * V;
        const float fh = (float)H, fw = (float)W;  // detect_net.py:163: x is divided by H, y by W
        float2 gc = {0.0f, 0.0f}, gw = {0.0f, 0.0f};
        if (g_center) gc = reinterpret_cast<const float2 *>(g_center)[o];
        if (g_wh) gw = reinterpret_cast<const float2 *>(g_wh)[o];
        const float go = g_obj ? g_obj[o] : 0.0f;
        const float s0 = yp_sigmoid(t[0]), s1 = yp_sigmoid(t[1]);
        gt[0] = (gc.x / fh) * ((1.0f - s0) * s0);
        gt[1] = (gc.y / fw) * ((1.0f - s1) * s1);
        const float w0 = (yp_exp(t[2]) * at.anc[2 * a + 0]) / fh;
        const float w1 = (yp_exp(t[3]) * at.anc[2 * a + 1]) / fw;
        gt[2] = gw.x * w0;
        gt[3] = gw.y * w1;
        const float so = yp_sigmoid(t[4]);
        gt[4] = go * ((1.0f - so) * so);
        float m = t[5];
        for (int j = 1; j < C; ++j) m = t[5 + j] > m ? t[5 + j] : m;
        float s = 0.0f;
        float pj[CS ? CS : 1];
        if (CS) {
#pragma unroll
            for (int j = 0; j < CS; ++j) pj[j] = yp_exp(t[5 + j] - m);
#pragma unroll
            for (int j = 0; j < CS; ++j) s = s + pj[j];
#pragma unroll
            for (int j = 0; j < CS; ++j) pj[j] = pj[j] / s;
        } else {
            for (int j = 0; j < C; ++j) s = s + yp_exp(t[5 + j] - m);
        }
        auto prob = [&](int j) { return CS ? pj[CS ? j : 0] : yp_exp(t[5 + j] - m) / s; };
        const float *gp = STAGED ? s_gcls + threadIdx.x * C : g_cls + o * C;
        float dot = 0.0f;
        if (g_cls) {
#pragma unroll
            for (int j = 0; j < (CS ? CS : C); ++j) dot = dot + gp[j] * prob(j);
        }
        if (STAGED) {
#pragma unroll
            for (int j = 0; j < kTileMaxWidth - 5; ++j)
                if (j < C) gt[5 + j] = g_cls ? (gp[j] - dot) * prob(j) : 0.0f;
        } else {
            float *out = g_pred + o * V;
            for (int v = 0; v < 5; ++v) out[v] = gt[v];
            for (int j = 0; j < C; ++j)
                out[5 + j] = g_cls ? (gp[j] - dot) * prob(j) : 0.0f;
        }
    }
    if (STAGED) {
        __syncthreads();   // every thread has read its raw row: the buffer now takes the results
        if (o < total) {
#pragma unroll
            for (int v = 0; v < kTileMaxWidth; ++v)
                if (v < V) s_tile[threadIdx.x * V + v] = gt[v];
        }
        yp_tile_store(g_pred + tile0 * V, (total - tile0) * V, V, s_tile);
    }
}

// ---- fused loss_custom: forward value and gradient of the raw head in one pass ------------
//
// train.py:123-197 restated per (cell, anchor) slot.  t = true grid vector, m = (t[0] != 0):
//   iou    = IoU of the predicted box (center, wh) and the "true" box (t[0:2], t[2:4])
//   det    = iou > iou_thr                         (a constant for the gradient)
//   conf   = noobj_w * (1-det) * (1-m) * obj^2 + obj_w * m * (1-obj)^2
//   class  = m * sum_j (onehot(int(t[4]))_j - cls_j)^2
//   coord  = coord_w * m * sum_k (t[k] - box[k])^2 ,  box = (center, wh)
//   total  = 0.5 * (sum conf + sum class + sum coord)
// The three sums are reduced per workgroup into `partials` and finished by k_loss_finish in a
// fixed order, so the value is reproducible run to run.

struct LossWeights {
    float coord_w, obj_w, noobj_w, iou_thr;
};

constexpr int kLossThreads = 256;

template <bool STAGED, int CS>   // CS as in k_decode_grid_ref_backward
__global__ void __launch_bounds__(kLossThreads)
k_loss_custom_ref(const float *__restrict__ pred, const float *__restrict__ truth, int H, int W, int A,
                  int C_arg, GradAnchors at, LossWeights lw, float *__restrict__ g_pred,
                  float *__restrict__ partials, long long total)
{
    // STAGED: [0, 256 V) the raw rows, later the gradient rows; then 256 x 5 target values
    extern __shared__ float s_tile[];
    __shared__ float s_red[3][kLossThreads / 64];
    static_assert(kLossThreads == kTileRows, "one record per thread");
    const long long tile0 = (long long)blockIdx.x * kTileRows;
    const long long o = tile0 + threadIdx.x;
    const int C = CS ? CS : C_arg;
    const int V = 5 + C;
    float *s_truth = s_tile + kTileRows * V;
    if (STAGED) {
        yp_tile_load(truth + tile0 * 5, (total - tile0) * 5, 5, s_truth);
        yp_tile_load(pred + tile0 * V, (total - tile0) * V, V, s_tile);
    }
    float l_conf = 0.0f, l_cls = 0.0f, l_coord = 0.0f;
    float gt[STAGED ? kTileMaxWidth : 5];
    if (o < total) {
        const int N = H * W * A;
        const int n = (int)(o % N);
        const int a = n % A, k = n / A;
        const float gx = (float)(k % H), gy = (float)(k / H);  // detect_net.py:155-162
        const float fh = (float)H, fw = (float)W;
        const float *t = STAGED ? s_tile + threadIdx.x * V : pred + o * V;
        const float *y = STAGED ? s_truth + threadIdx.x * 5 : truth + o * 5;
        const float s0 = yp_sigmoid(t[0]), s1 = yp_sigmoid(t[1]);
        const float c0 = (s0 + gx) / fh, c1 = (s1 + gy) / fw;
        const float w0 = (yp_exp(t[2]) * at.anc[2 * a + 0]) / fh;
        const float w1 = (yp_exp(t[3]) * at.anc[2 * a + 1]) / fw;
        const float ob = yp_sigmoid(t[4]);
        const float y0 = y[0], y1 = y[1], y2 = y[2], y3 = y[3];
        const float m = (y0 != 0.0f) ? 1.0f : 0.0f;  // create_mask(...)[..., :1], train.py:110-114
        // IoU of prediction and target, train.py:132-157
        const float pmin0 = c0 - w0 / 2.0f, pmin1 = c1 - w1 / 2.0f;
        const float pmax0 = c0 + w0 / 2.0f, pmax1 = c1 + w1 / 2.0f;
        const float tmin0 = y0 - y2 / 2.0f, tmin1 = y1 - y3 / 2.0f;
        const float tmax0 = y0 + y2 / 2.0f, tmax1 = y1 + y3 / 2.0f;
        float i0 = (pmax0 < tmax0 ? pmax0 : tmax0) - (pmin0 > tmin0 ? pmin0 : tmin0);
        float i1 = (pmax1 < tmax1 ? pmax1 : tmax1) - (pmin1 > tmin1 ? pmin1 : tmin1);
        i0 = i0 > 0.0f ? i0 : 0.0f;
        i1 = i1 > 0.0f ? i1 : 0.0f;
        const float inter = i0 * i1;
        const float iou = inter / (w0 * w1 + y2 * y3 - inter);
        const float det = (iou > lw.iou_thr) ? 1.0f : 0.0f;  // :167
        const float noobj = lw.noobj_w * (1.0f - det) * (1.0f - m);  // :170
        l_conf = noobj * (ob * ob) + lw.obj_w * m * ((1.0f - ob) * (1.0f - ob));  // :171-175
        const float d0 = y0 - c0, d1 = y1 - c1, d2 = y2 - w0, d3 = y3 - w1;
        l_coord = lw.coord_w * m * (d0 * d0) + lw.coord_w * m * (d1 * d1) +
                  lw.coord_w * m * (d2 * d2) + lw.coord_w * m * (d3 * d3);  // :188
        // softmax class probabilities, detect_net.py:180
        float mx = t[5];
        for (int j = 1; j < C; ++j) mx = t[5 + j] > mx ? t[5 + j] : mx;
        float se = 0.0f;
        float pj[CS ? CS : 1];
        if (CS) {
#pragma unroll
            for (int j = 0; j < CS; ++j) pj[j] = yp_exp(t[5 + j] - mx);
#pragma unroll
            for (int j = 0; j < CS; ++j) se = se + pj[j];
#pragma unroll
            for (int j = 0; j < CS; ++j) pj[j] = pj[j] / se;
        } else {
            for (int j = 0; j < C; ++j) se = se + yp_exp(t[5 + j] - mx);
        }
        auto prob = [&](int j) { return CS ? pj[CS ? j : 0] : yp_exp(t[5 + j] - mx) / se; };
        const int cls_id = (int)y[4];  // tf.cast(..., tf.int32), train.py:177
        float dot = 0.0f;              // sum_j (dL/dp_j) * p_j for the softmax backward
#pragma unroll
        for (int j = 0; j < (CS ? CS : C); ++j) {
            const float p = prob(j);
            const float e = ((j == cls_id) ? 1.0f : 0.0f) - p;
            l_cls = l_cls + m * (e * e);  // :179
            dot = dot + (-(m * e)) * p;
        }
        if (g_pred) {
            // d total / d output (the 0.5 cancels the 2 of each square)
            const float g_c0 = -(lw.coord_w * m * d0), g_c1 = -(lw.coord_w * m * d1);
            const float g_w0 = -(lw.coord_w * m * d2), g_w1 = -(lw.coord_w * m * d3);
            const float g_ob = noobj * ob - lw.obj_w * m * (1.0f - ob);
            gt[0] = (g_c0 / fh) * ((1.0f - s0) * s0);
            gt[1] = (g_c1 / fw) * ((1.0f - s1) * s1);
            gt[2] = g_w0 * w0;
            gt[3] = g_w1 * w1;
            gt[4] = g_ob * ((1.0f - ob) * ob);
            if (STAGED) {
#pragma unroll
                for (int j = 0; j < kTileMaxWidth - 5; ++j) {
                    if (j < C) {
                        const float p = prob(j);
                        const float e = ((j == cls_id) ? 1.0f : 0.0f) - p;
                        gt[5 + j] = (-(m * e) - dot) * p;
                    }
                }
            } else {
                float *out = g_pred + o * V;
                for (int v = 0; v < 5; ++v) out[v] = gt[v];
                for (int j = 0; j < C; ++j) {
                    const float p = prob(j);
                    const float e = ((j == cls_id) ? 1.0f : 0.0f) - p;
                    out[5 + j] = (-(m * e) - dot) * p;
                }
            }
        }
    }
    if (STAGED && g_pred) {
        __syncthreads();   // every thread has read its raw row: the buffer now takes the results
        if (o < total) {
#pragma unroll
            for (int v = 0; v < kTileMaxWidth; ++v)
                if (v < V) s_tile[threadIdx.x * V + v] = gt[v];
        }
        yp_tile_store(g_pred + tile0 * V, (total - tile0) * V, V, s_tile);
    }
    float v[3] = {l_conf, l_cls, l_coord};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        float x = v[q];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
        if ((threadIdx.x & 63) == 0) s_red[q][threadIdx.x >> 6] = x;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        float x = 0.0f;
        for (int w = 0; w < kLossThreads / 64; ++w) x += s_red[threadIdx.x][w];
        partials[(size_t)blockIdx.x * 3 + threadIdx.x] = x;
    }
}

// One workgroup adds the per-workgroup partial sums in binary64, in a fixed order.
constexpr int kFinishThreads = 1024;

__global__ void __launch_bounds__(kFinishThreads)
k_loss_finish(const float *__restrict__ partials, int blocks, float *__restrict__ loss)
{
    __shared__ double s_red[3][kFinishThreads / 64];
    double v[3] = {0.0, 0.0, 0.0};
    for (int i = (int)threadIdx.x; i < blocks; i += kFinishThreads)
        for (int q = 0; q < 3; ++q) v[q] += (double)partials[(size_t)i * 3 + q];
    for (int q = 0; q < 3; ++q) {
        double x = v[q];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
        if ((threadIdx.x & 63) == 0) s_red[q][threadIdx.x >> 6] = x;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double sum[3];
        for (int q = 0; q < 3; ++q) {
            sum[q] = 0.0;
            for (int w = 0; w < kFinishThreads / 64; ++w) sum[q] += s_red[q][w];
        }
        loss[0] = (float)(0.5 * (sum[0] + sum[1] + sum[2]));  // train.py:195
        loss[1] = (float)sum[0];
        loss[2] = (float)sum[1];
        loss[3] = (float)sum[2];
    }
}

}  // namespace

extern "C" size_t yp_loss_custom_reference_workspace_bytes(int batch, int grid_h, int grid_w,
                                                           int num_anchors)
{
    if (batch <= 0 || grid_h <= 0 || grid_w <= 0 || num_anchors <= 0) return 0;
    const long long total = (long long)batch * grid_h * grid_w * num_anchors;
    const size_t blocks = (size_t)((total + kLossThreads - 1) / kLossThreads);
    return (blocks * 3 * sizeof(float) + 255) & ~(size_t)255;
}

extern "C" int yp_loss_custom_reference(const float *d_pred, const float *d_true_grid, int batch,
                                        int grid_h, int grid_w, int num_anchors, int num_classes,
                                        const float *anchors, float coord_weight, float obj_weight,
                                        float noobj_weight, float iou_thr, float *d_loss,
                                        float *d_grad_pred, void *d_workspace,
                                        size_t workspace_bytes, void *stream)
{
    if (!d_pred || !d_true_grid || !d_loss || !anchors || batch <= 0 || grid_h <= 0 ||
        grid_w <= 0 || num_anchors <= 0 || num_classes <= 0)
        return YP_ERR_ARG;
    if (num_anchors > YP_MAX_ANCHORS || num_classes > kGradMaxClasses) return YP_ERR_UNSUPPORTED;
    const long long total = (long long)batch * grid_h * grid_w * num_anchors;
    const long long blocks = (total + kLossThreads - 1) / kLossThreads;
    if (blocks > 0x3fffffffLL) return YP_ERR_UNSUPPORTED;
    if (!d_workspace || ((uintptr_t)d_workspace & 255) ||
        workspace_bytes < yp_loss_custom_reference_workspace_bytes(batch, grid_h, grid_w, num_anchors))
        return YP_ERR_WORKSPACE;
    GradAnchors at;
    for (int k = 0; k < 2 * YP_MAX_ANCHORS; ++k) at.anc[k] = k < 2 * num_anchors ? anchors[k] : 0.0f;
    const LossWeights lw = {coord_weight, obj_weight, noobj_weight, iou_thr};
    float *partials = (float *)d_workspace;
    const int V = 5 + num_classes;
    if (num_classes == 5)
        hipLaunchKernelGGL((k_loss_custom_ref<true, 5>), dim3((unsigned)blocks), dim3(kLossThreads),
                           (size_t)kTileRows * (V + 5) * sizeof(float), (hipStream_t)stream, d_pred,
                           d_true_grid, grid_h, grid_w, num_anchors, num_classes, at, lw, d_grad_pred,
                           partials, total);
    else if (V <= kTileMaxWidth)
        hipLaunchKernelGGL((k_loss_custom_ref<true, 0>), dim3((unsigned)blocks), dim3(kLossThreads),
                           (size_t)kTileRows * (V + 5) * sizeof(float), (hipStream_t)stream, d_pred,
                           d_true_grid, grid_h, grid_w, num_anchors, num_classes, at, lw, d_grad_pred,
                           partials, total);
    else
        hipLaunchKernelGGL((k_loss_custom_ref<false, 0>), dim3((unsigned)blocks), dim3(kLossThreads), 0,
                           (hipStream_t)stream, d_pred, d_true_grid, grid_h, grid_w, num_anchors,
                           num_classes, at, lw, d_grad_pred, partials, total);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_loss_finish, dim3(1), dim3(kFinishThreads), 0, (hipStream_t)stream, partials,
                       (int)blocks, d_loss);
    return (int)hipGetLastError();
}

extern "C" int yp_predict_transform_reference_backward(
    const float *d_pred, int batch, int grid_h, int grid_w, int num_anchors, int num_classes,
    const float *anchors, const float *d_grad_center, const float *d_grad_wh,
    const float *d_grad_obj, const float *d_grad_cls, float *d_grad_pred, void *stream)
{
    if (!d_pred || !d_grad_pred || !anchors || batch <= 0 || grid_h <= 0 || grid_w <= 0 ||
        num_anchors <= 0 || num_classes <= 0)
        return YP_ERR_ARG;
    if (num_anchors > YP_MAX_ANCHORS || num_classes > kGradMaxClasses) return YP_ERR_UNSUPPORTED;
    GradAnchors at;
    for (int k = 0; k < 2 * YP_MAX_ANCHORS; ++k) at.anc[k] = k < 2 * num_anchors ? anchors[k] : 0.0f;
    const long long total = (long long)batch * grid_h * grid_w * num_anchors;
    const long long blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return YP_ERR_UNSUPPORTED;
    const int V = 5 + num_classes;
    if (num_classes == 5)
        hipLaunchKernelGGL((k_decode_grid_ref_backward<true, 5>), dim3((unsigned)blocks), dim3(kTileRows),
                           (size_t)kTileRows * (V + num_classes) * sizeof(float), (hipStream_t)stream,
                           d_pred, grid_h, grid_w, num_anchors, num_classes, at, d_grad_center,
                           d_grad_wh, d_grad_obj, d_grad_cls, d_grad_pred, total);
    else if (V <= kTileMaxWidth)
        hipLaunchKernelGGL((k_decode_grid_ref_backward<true, 0>), dim3((unsigned)blocks), dim3(kTileRows),
                           (size_t)kTileRows * (V + num_classes) * sizeof(float), (hipStream_t)stream,
                           d_pred, grid_h, grid_w, num_anchors, num_classes, at, d_grad_center,
                           d_grad_wh, d_grad_obj, d_grad_cls, d_grad_pred, total);
    else
        hipLaunchKernelGGL((k_decode_grid_ref_backward<false, 0>), dim3((unsigned)blocks), dim3(kTileRows), 0,
                           (hipStream_t)stream, d_pred, grid_h, grid_w, num_anchors, num_classes, at,
                           d_grad_center, d_grad_wh, d_grad_obj, d_grad_cls, d_grad_pred, total);
    return (int)hipGetLastError();
}
