// metrics.hip -- the validation statistics of libyolopost.so for MI355X (gfx950): kept
// detections against the ground-truth boxes of the same images.
//
//   yp_match_detections   detection rows (the (D, 8) list write_results emits) x decoded
//                         ground truth (the output of yp_decode_true_box_grid) ->
//                         per detection: the ground-truth row it claims, or why it is a false
//                         positive; per class: true positives, false positives, misses,
//                         duplicates, localised-but-wrong-class
//
// This is the comparison train.py:365-366 describes ("check iou of each nms box with each
// ground truth from val set, if above threshold compare classification") and leaves to the
// empty comp_nms_gt (train.py:279-280); get_metrics (train.py:331-340) only prints the two
// location lists.  Rule (the PASCAL-VOC one): detections are taken in list order, which the
// detection path emits by descending confidence within an image and class; a detection's
// candidate is the same-image, same-class ground-truth box of highest IoU (lowest row on equal
// IoU); it is a true positive when that IoU reaches the threshold and no earlier detection
// claimed the same box, otherwise a false positive.  "First claim wins" needs no sequential
// loop: every detection finds its candidate independently and an atomic minimum over list
// positions per ground-truth box names the winner, so the result does not depend on the
// order threads run in.
//
// IoU here is the plain geometric one (no +1 pixel term; the ground-truth boxes are real-valued
// centre/size boxes), evaluated in binary32 in a fixed operation order with a correctly rounded
// divide, so that a CPU restatement reproduces every comparison bit for bit.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "yolopost.h"

namespace {

constexpr int kMatchThreads = 256;
constexpr int kUnclaimed = 0x7f7f7f7f;   // what hipMemsetAsync(.., 0x7f, ..) leaves in an int32
constexpr int kHistMax = 2048;           // class-count cells kept in LDS per workgroup

struct MatchArgs {
    const float *det_rows;     // (det_cap, 8): image, x1, y1, x2, y2, objectness, class score, class id
    const int32_t *det_count;
    int64_t det_cap;
    const float *gt_boxes;     // (gt_cap, 4): cx, cy, w, h
    const float *gt_class;     // (gt_cap,)
    const int32_t *gt_index;   // (gt_cap, 4): image, row, col, anchor; ascending by image
    const int32_t *gt_count;
    int64_t gt_cap;
    int32_t num_classes;
    float iou_thr;
    int32_t *det_match;        // (det_cap,)
    float *det_iou;            // (det_cap,) or null
    int32_t *class_counts;     // (YP_MATCH_PLANES, C)
    int32_t *claimed;          // workspace (gt_cap,): list position of the first claimant
};

__device__ __forceinline__ int clamp_count(const int32_t *p, int64_t cap)
{
    const int64_t n = (int64_t)*p;
    return (int)(n < 0 ? 0 : (n > cap ? cap : n));
}

// IoU of corners d with the centre/size box g; 0 when the union is not positive.
__device__ __forceinline__ float match_iou(const float4 d, const float4 g)
{
    const float hw = g.z * 0.5f, hh = g.w * 0.5f;
    const float gx1 = g.x - hw, gy1 = g.y - hh, gx2 = g.x + hw, gy2 = g.y + hh;
    const float ix = fmaxf(fminf(d.z, gx2) - fmaxf(d.x, gx1), 0.0f);
    const float iy = fmaxf(fminf(d.w, gy2) - fmaxf(d.y, gy1), 0.0f);
    const float inter = ix * iy;
    const float area_d = (d.z - d.x) * (d.w - d.y);
    const float area_g = (gx2 - gx1) * (gy2 - gy1);
    const float uni = (area_d + area_g) - inter;
    return uni > 0.0f ? inter / uni : 0.0f;
}

// Pass 1, one thread per detection: the candidate ground-truth box and the claim on it.
// det_match[d] receives the candidate row, or YP_MATCH_NONE / YP_MATCH_OTHER_CLASS.
__global__ void __launch_bounds__(kMatchThreads) k_match_claim(MatchArgs a)
{
    const int n_det = clamp_count(a.det_count, a.det_cap);
    const int n_gt = clamp_count(a.gt_count, a.gt_cap);
    const int64_t d = (int64_t)blockIdx.x * kMatchThreads + threadIdx.x;
    if (d >= n_det) return;
    const float4 r0 = *reinterpret_cast<const float4 *>(a.det_rows + d * 8);
    const float4 r1 = *reinterpret_cast<const float4 *>(a.det_rows + d * 8 + 4);
    const int img = (int)r0.x;
    const float4 box = make_float4(r0.y, r0.z, r0.w, r1.x);
    const float cls = r1.w;
    // [lo, hi): the ground-truth rows of this image (the index is ascending by image)
    int lo = 0, hi = n_gt;
    while (lo < hi) {
        const int m = (lo + hi) >> 1;
        if (a.gt_index[(int64_t)m * 4] < img) lo = m + 1; else hi = m;
    }
    const int first = lo;
    hi = n_gt;
    while (lo < hi) {
        const int m = (lo + hi) >> 1;
        if (a.gt_index[(int64_t)m * 4] <= img) lo = m + 1; else hi = m;
    }
    const int last = lo;
    float best_any = 0.0f, best_same = 0.0f, best_other = 0.0f;
    int cand = -1;
    for (int g = first; g < last; ++g) {
        const float4 gb = *reinterpret_cast<const float4 *>(a.gt_boxes + (int64_t)g * 4);
        const float iou = match_iou(box, gb);
        if (iou > best_any) best_any = iou;
        if (a.gt_class[g] == cls) {
            if (iou > best_same) { best_same = iou; cand = g; }
        } else if (iou > best_other) {
            best_other = iou;
        }
    }
    int verdict;
    if (cand >= 0 && best_same >= a.iou_thr) {
        verdict = cand;
        atomicMin(&a.claimed[cand], (int)d);
    } else if (best_other >= a.iou_thr) {
        verdict = YP_MATCH_OTHER_CLASS;
    } else {
        verdict = YP_MATCH_NONE;
    }
    a.det_match[d] = verdict;
    if (a.det_iou) a.det_iou[d] = best_any;
}

// Pass 2, one thread per detection and per ground-truth box: settle the claims, count per class.
__global__ void __launch_bounds__(kMatchThreads) k_match_count(MatchArgs a)
{
    __shared__ int32_t s_hist[kHistMax];
    const int C = a.num_classes;
    const int cells = YP_MATCH_PLANES * C;
    const bool in_lds = cells <= kHistMax;
    if (in_lds) {
        for (int e = (int)threadIdx.x; e < cells; e += kMatchThreads) s_hist[e] = 0;
        __syncthreads();
    }
    auto bump = [&](int plane, float cls) {
        if (!(cls >= 0.0f && cls < (float)C)) return;       // class ids outside [0, C) are not counted
        const int cell = plane * C + (int)cls;
        if (in_lds) atomicAdd(&s_hist[cell], 1); else atomicAdd(&a.class_counts[cell], 1);
    };
    const int n_det = clamp_count(a.det_count, a.det_cap);
    const int n_gt = clamp_count(a.gt_count, a.gt_cap);
    const int64_t t = (int64_t)blockIdx.x * kMatchThreads + threadIdx.x;
    if (t < n_det) {
        const float cls = a.det_rows[t * 8 + 7];
        int v = a.det_match[t];
        if (v >= 0 && v < n_gt) {
            if (a.claimed[v] == (int)t) {
                bump(0, cls);                                // true positive
            } else {
                v = YP_MATCH_DUPLICATE;
                a.det_match[t] = v;
            }
        }
        if (v < 0) {
            bump(1, cls);                                    // false positive, of any kind
            if (v == YP_MATCH_DUPLICATE) bump(3, cls);
            if (v == YP_MATCH_OTHER_CLASS) bump(4, cls);
        }
    }
    if (t < n_gt && a.claimed[t] == kUnclaimed) bump(2, a.gt_class[t]);   // miss
    if (in_lds) {
        __syncthreads();
        for (int e = (int)threadIdx.x; e < cells; e += kMatchThreads) {
            const int v = s_hist[e];
            if (v) atomicAdd(&a.class_counts[e], v);
        }
    }
}

size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

}  // namespace

extern "C" {

size_t yp_match_detections_workspace_bytes(int64_t gt_cap)
{
    if (gt_cap <= 0 || gt_cap > ((int64_t)1 << 30)) return 0;
    return align256((size_t)gt_cap * sizeof(int32_t));
}

int yp_match_detections(const float *d_det_rows, const int32_t *d_det_count, int64_t det_cap,
                        const float *d_gt_boxes, const float *d_gt_class,
                        const int32_t *d_gt_index, const int32_t *d_gt_count, int64_t gt_cap,
                        int num_classes, float iou_thr, int32_t *d_det_match, float *d_det_iou,
                        int32_t *d_class_counts, void *d_workspace, size_t workspace_bytes,
                        void *stream)
{
    if (!d_det_rows || !d_det_count || !d_gt_boxes || !d_gt_class || !d_gt_index || !d_gt_count ||
        !d_det_match || !d_class_counts)
        return YP_ERR_ARG;
    if (det_cap <= 0 || det_cap > ((int64_t)1 << 30) || gt_cap <= 0 || gt_cap > ((int64_t)1 << 30))
        return YP_ERR_ARG;
    if (num_classes <= 0 || num_classes > 4096) return YP_ERR_ARG;
    if (!(iou_thr > 0.0f) || !(iou_thr <= 1.0f)) return YP_ERR_ARG;     // (rejects NaN too)
    if (((uintptr_t)d_det_rows & 15) || ((uintptr_t)d_gt_boxes & 15)) return YP_ERR_ARG;
    if (!d_workspace || ((uintptr_t)d_workspace & 255) ||
        workspace_bytes < yp_match_detections_workspace_bytes(gt_cap))
        return YP_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    MatchArgs a;
    a.det_rows = d_det_rows; a.det_count = d_det_count; a.det_cap = det_cap;
    a.gt_boxes = d_gt_boxes; a.gt_class = d_gt_class; a.gt_index = d_gt_index;
    a.gt_count = d_gt_count; a.gt_cap = gt_cap;
    a.num_classes = num_classes; a.iou_thr = iou_thr;
    a.det_match = d_det_match; a.det_iou = d_det_iou; a.class_counts = d_class_counts;
    a.claimed = (int32_t *)d_workspace;
    hipError_t e = hipMemsetAsync(a.claimed, 0x7f, (size_t)gt_cap * sizeof(int32_t), st);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(d_class_counts, 0, (size_t)YP_MATCH_PLANES * num_classes * sizeof(int32_t), st);
    if (e != hipSuccess) return (int)e;
    const unsigned det_blocks = (unsigned)((det_cap + kMatchThreads - 1) / kMatchThreads);
    const int64_t both = det_cap > gt_cap ? det_cap : gt_cap;
    const unsigned all_blocks = (unsigned)((both + kMatchThreads - 1) / kMatchThreads);
    hipLaunchKernelGGL(k_match_claim, dim3(det_blocks), dim3(kMatchThreads), 0, st, a);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_match_count, dim3(all_blocks), dim3(kMatchThreads), 0, st, a);
    return (int)hipGetLastError();
}

}  // extern "C"
