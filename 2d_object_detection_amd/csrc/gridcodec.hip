// gridcodec.hip -- the ground-truth grid codec of libyolopost.so for MI355X (gfx950): the data
// format on the label side of the detection path.
//
//   yp_encode_true_box_grid   objects [cx, cy, w, h, class] (float64, original pixels) ->
//                             (B, H, W, A, 5) float32 target grid + (B, H, W, A, 2) mask
//                             (create_true_box_grid, create_dataset.py:47-115)
//   yp_decode_true_box_grid   target grid -> occupied (image, row, col, anchor) locations in
//                             sorted order, their boxes in original pixels and class ids
//                             (gather_grid_boxes, create_dataset.py:294-328; the location list
//                             is also what get_metrics builds, train.py:332-338)
//
// The encoder works in binary64 exactly as the numpy code does (float32 anchors promoted
// where numpy promotes them); the decoder works in binary32 with the library's fixed exp.
// Both are small byte-moving kernels: one workgroup per image, ordered wave-ballot
// compaction, no atomics, so the output order is deterministic.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "yolopost.h"
#include "yp_math.h"

namespace {

constexpr int kCodecThreads = 256;
constexpr int kCodecWaves = kCodecThreads / 64;

struct GridGeom {
    int32_t batch, gh, gw, na;
    int32_t im_h, im_w;
    float anchors[2 * YP_MAX_ANCHORS];
};

__device__ __forceinline__ int lanes_below_mask(uint64_t m)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                          __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// One workgroup per image.  The image's grid and mask are cleared, then its objects are
// handled in chunks of kCodecThreads in list order; inside a chunk an object is stored only
// if no later object of the chunk claims the same (row, col, anchor) slot, so that the last
// object of the list wins a slot exactly as the sequential loop of the reference does.
__global__ void __launch_bounds__(kCodecThreads)
k_encode_true_grid(const double *__restrict__ objects, const int32_t *__restrict__ offsets,
                   GridGeom g, float *__restrict__ grid, float *__restrict__ mask)
{
    __shared__ int32_t s_slot[kCodecThreads];
    const int b = (int)blockIdx.x;
    const int slots = g.gh * g.gw * g.na;
    float *gimg = grid + (size_t)b * slots * 5;
    float *mimg = mask + (size_t)b * slots * 2;
    for (int e = (int)threadIdx.x; e < slots * 5; e += kCodecThreads) gimg[e] = 0.0f;
    for (int e = (int)threadIdx.x; e < slots * 2; e += kCodecThreads) mimg[e] = 0.0f;
    const int first = offsets[b], last = offsets[b + 1];
    for (int base = first; base < last; base += kCodecThreads) {
        __threadfence_block();
        __syncthreads();  // clears / the previous chunk's stores are ordered before ours
        const int o = base + (int)threadIdx.x;
        int slot = -1;
        float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, v3 = 0.0f, v4 = 0.0f;
        if (o < last) {
            const double *y = objects + (size_t)o * 5;
            // create_dataset.py:56-64: to fractions of the image, then to grid units
            const double cx = y[0] / (double)g.im_w * (double)g.gw;
            const double cy = y[1] / (double)g.im_h * (double)g.gh;
            const double bw = y[2] / (double)g.im_w * (double)g.gw;
            const double bh = y[3] / (double)g.im_h * (double)g.gh;
            const double fi = floor(cy), fj = floor(cx);  // :74-75 row, column
            double best_iou = 0.0;
            int best = 0;
            const double bmax0 = bw / 2, bmax1 = bh / 2;
            for (int k = 0; k < g.na; ++k) {  // :79-98
                const float aw = g.anchors[2 * k], ah = g.anchors[2 * k + 1];
                const double amax0 = (double)(aw / 2), amax1 = (double)(ah / 2);
                const double mn0 = -bmax0 > -amax0 ? -bmax0 : -amax0;
                const double mn1 = -bmax1 > -amax1 ? -bmax1 : -amax1;
                const double mx0 = bmax0 < amax0 ? bmax0 : amax0;
                const double mx1 = bmax1 < amax1 ? bmax1 : amax1;
                double w0 = mx0 - mn0, w1 = mx1 - mn1;
                w0 = w0 > 0.0 ? w0 : 0.0;
                w1 = w1 > 0.0 ? w1 : 0.0;
                const double inter = w0 * w1;
                const double box_area = bw * bh;
                const double anchor_area = (double)(aw * ah);  // float32 product, as numpy
                const double iou = inter / (box_area + anchor_area - inter);
                if (iou > best_iou) {
                    best_iou = iou;
                    best = k;
                }
            }
            const bool inside = fi >= 0.0 && fi < (double)g.gh && fj >= 0.0 && fj < (double)g.gw;
            if (best_iou > 0.0 && inside) {  // :100
                const int i = (int)fi, j = (int)fj;
                slot = (i * g.gw + j) * g.na + best;
                v0 = (float)(cx - (double)j);
                v1 = (float)(cy - (double)i);
                v2 = (float)log(bw / (double)g.anchors[2 * best]);
                v3 = (float)log(bh / (double)g.anchors[2 * best + 1]);
                v4 = (float)y[4];
            }
        }
        s_slot[threadIdx.x] = slot;
        __syncthreads();
        if (slot >= 0) {
            const int n = min(kCodecThreads, last - base);
            bool overwritten = false;
            for (int t = (int)threadIdx.x + 1; t < n; ++t) overwritten |= (s_slot[t] == slot);
            if (!overwritten) {
                float *dst = gimg + (size_t)slot * 5;
                dst[0] = v0; dst[1] = v1; dst[2] = v2; dst[3] = v3; dst[4] = v4;
                mimg[(size_t)slot * 2] = 1.0f;
                mimg[(size_t)slot * 2 + 1] = 1.0f;
            }
        }
    }
}

// Pass 1 of the decoder: per image, the occupied slots (any of the 5 values != 0) in
// ascending slot order into slot_list[b*slots + r], and their number into counts[b].
__global__ void __launch_bounds__(kCodecThreads)
k_grid_locate(const float *__restrict__ grid, GridGeom g, int32_t *__restrict__ slot_list,
              int32_t *__restrict__ counts)
{
    __shared__ int s_wave[kCodecWaves];
    const int b = (int)blockIdx.x;
    const int slots = g.gh * g.gw * g.na;
    const float *gimg = grid + (size_t)b * slots * 5;
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    int running = 0;
    for (int base = 0; base < slots; base += kCodecThreads) {
        const int s = base + (int)threadIdx.x;
        bool occ = false;
        if (s < slots) {
            const float *v = gimg + (size_t)s * 5;
            occ = (v[0] != 0.0f) | (v[1] != 0.0f) | (v[2] != 0.0f) | (v[3] != 0.0f) | (v[4] != 0.0f);
        }
        const uint64_t m = __ballot(occ);
        if (lane == 0) s_wave[wave] = __popcll(m);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kCodecWaves; ++w) {
            const int c = s_wave[w];
            before += (w < wave) ? c : 0;
            total += c;
        }
        if (occ) slot_list[(size_t)b * slots + running + before + lanes_below_mask(m)] = s;
        running += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) counts[b] = running;
}

// Pass 2: each image's list is decoded to its place in the batch-wide output.
__global__ void __launch_bounds__(kCodecThreads)
k_grid_emit(const float *__restrict__ grid, GridGeom g, const int32_t *__restrict__ slot_list,
            const int32_t *__restrict__ counts, int32_t *__restrict__ out_index,
            float *__restrict__ out_boxes, float *__restrict__ out_class, int64_t cap,
            int32_t *__restrict__ out_count)
{
    __shared__ int s_part[kCodecWaves];
    const int b = (int)blockIdx.x;
    const int slots = g.gh * g.gw * g.na;
    int part = 0;
    for (int i = (int)threadIdx.x; i < b; i += kCodecThreads) part += counts[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = part;
    __syncthreads();
    int64_t start = 0;
#pragma unroll
    for (int w = 0; w < kCodecWaves; ++w) start += s_part[w];
    const int n = counts[b];
    if (b == g.batch - 1 && threadIdx.x == 0) *out_count = (int32_t)(start + n);
    const float *gimg = grid + (size_t)b * slots * 5;
    const float conv_w = (float)g.gw, conv_h = (float)g.gh;
    const float orig_w = (float)g.im_w, orig_h = (float)g.im_h;
    for (int r = (int)threadIdx.x; r < n; r += kCodecThreads) {
        const int64_t row = start + r;
        if (row >= cap) break;
        const int s = slot_list[(size_t)b * slots + r];
        const int a = s % g.na, cell = s / g.na;
        const int j = cell % g.gw, i = cell / g.gw;
        const float *v = gimg + (size_t)s * 5;
        const float cx = v[0] + (float)j;  // create_dataset.py:306-318: x offset = column
        const float cy = v[1] + (float)i;
        const float w = yp_exp(v[2]) * g.anchors[2 * a];      // :320-322
        const float h = yp_exp(v[3]) * g.anchors[2 * a + 1];
        int32_t *oi = out_index + row * 4;
        oi[0] = b; oi[1] = i; oi[2] = j; oi[3] = a;
        float *ob = out_boxes + row * 4;
        ob[0] = (cx / conv_w) * orig_w;  // :327-328
        ob[1] = (cy / conv_h) * orig_h;
        ob[2] = (w / conv_w) * orig_w;
        ob[3] = (h / conv_h) * orig_h;
        out_class[row] = v[4];
    }
}

int make_geom(GridGeom &g, int batch, int gh, int gw, int na, const float *anchors, int im_h,
              int im_w)
{
    if (batch <= 0 || gh <= 0 || gw <= 0 || na <= 0 || !anchors || im_h <= 0 || im_w <= 0)
        return YP_ERR_ARG;
    if (na > YP_MAX_ANCHORS) return YP_ERR_UNSUPPORTED;
    if ((int64_t)gh * gw * na * 5 > (int64_t)1 << 30) return YP_ERR_UNSUPPORTED;
    g.batch = batch; g.gh = gh; g.gw = gw; g.na = na; g.im_h = im_h; g.im_w = im_w;
    for (int k = 0; k < 2 * YP_MAX_ANCHORS; ++k) g.anchors[k] = k < 2 * na ? anchors[k] : 0.0f;
    return YP_OK;
}

size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

}  // namespace

extern "C" {

int yp_encode_true_box_grid(const double *d_objects, const int32_t *d_object_offsets, int batch,
                            int grid_h, int grid_w, int num_anchors, const float *anchors,
                            int im_h, int im_w, float *d_grid, float *d_mask, void *stream)
{
    GridGeom g;
    const int rc = make_geom(g, batch, grid_h, grid_w, num_anchors, anchors, im_h, im_w);
    if (rc != YP_OK) return rc;
    if (!d_object_offsets || !d_grid || !d_mask) return YP_ERR_ARG;
    hipLaunchKernelGGL(k_encode_true_grid, dim3((unsigned)batch), dim3(kCodecThreads), 0,
                       (hipStream_t)stream, d_objects, d_object_offsets, g, d_grid, d_mask);
    return (int)hipGetLastError();
}

size_t yp_true_box_grid_workspace_bytes(int batch, int grid_h, int grid_w, int num_anchors)
{
    if (batch <= 0 || grid_h <= 0 || grid_w <= 0 || num_anchors <= 0) return 0;
    return align256((size_t)batch * grid_h * grid_w * num_anchors * sizeof(int32_t)) +
           align256((size_t)batch * sizeof(int32_t));
}

int yp_decode_true_box_grid(const float *d_grid, int batch, int grid_h, int grid_w,
                            int num_anchors, const float *anchors, int im_h, int im_w,
                            int32_t *d_out_index, float *d_out_boxes, float *d_out_class,
                            int64_t out_cap, int32_t *d_out_count, void *d_workspace,
                            size_t workspace_bytes, void *stream)
{
    GridGeom g;
    const int rc = make_geom(g, batch, grid_h, grid_w, num_anchors, anchors, im_h, im_w);
    if (rc != YP_OK) return rc;
    if (!d_grid || !d_out_index || !d_out_boxes || !d_out_class || !d_out_count || out_cap < 0)
        return YP_ERR_ARG;
    if (!d_workspace || ((uintptr_t)d_workspace & 255) ||
        workspace_bytes < yp_true_box_grid_workspace_bytes(batch, grid_h, grid_w, num_anchors))
        return YP_ERR_WORKSPACE;
    int32_t *slot_list = (int32_t *)d_workspace;
    int32_t *counts = (int32_t *)((char *)d_workspace +
                                  align256((size_t)batch * grid_h * grid_w * num_anchors *
                                           sizeof(int32_t)));
    hipLaunchKernelGGL(k_grid_locate, dim3((unsigned)batch), dim3(kCodecThreads), 0,
                       (hipStream_t)stream, d_grid, g, slot_list, counts);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_grid_emit, dim3((unsigned)batch), dim3(kCodecThreads), 0,
                       (hipStream_t)stream, d_grid, g, slot_list, counts, d_out_index,
                       d_out_boxes, d_out_class, out_cap, d_out_count);
    return (int)hipGetLastError();
}

}  // extern "C"
