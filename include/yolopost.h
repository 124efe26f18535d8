/*
 * yolopost.h -- C ABI of libyolopost.so, the MI355X (gfx950) implementation of the YOLO
 * detection post-processing path: anchor decode -> confidence filter -> greedy NMS.
 *
 * The reference (lbadams2/2D_object_detection) has no native or FFI layer: the path is a
 * Python class of static methods, DetectNet in detect_net.py:75-183, called from
 * train.py:128, :205 and :347-348.  Each entry point below names the Python function it
 * replaces.  Two semantic modes exist:
 *
 *   YP_MODE_REFERENCE  detect_net.py exactly: one H x W grid, A anchors in grid units,
 *                      memory order (b,row,col,anchor,attr), softmax class probabilities,
 *                      score = objectness * class prob, filter score >= conf, boxes
 *                      [y1,x1,y2,x2] scaled to (im_h, im_w), class-agnostic NMS that
 *                      suppresses IoU > thr and keeps at most max_out boxes per image.
 *   YP_MODE_YOLOV3     the three-stride YOLOv3 post-processing of the tutorial credited in
 *                      README.md:1 (predict_transform / write_results / bbox_iou): NCHW
 *                      heads, sigmoid class scores, filter objectness > conf, boxes
 *                      [x1,y1,x2,y2] in input pixels, per-image per-class NMS ordered by
 *                      objectness that keeps a later box only if IoU(+1 form) < thr.
 *                      These functions are not present in the reference repository.
 *
 * Conventions
 *   - every pointer named d_* or inside YpHead.data is DEVICE memory owned by the caller;
 *     the library never allocates or frees device memory and never synchronises: kernels
 *     are enqueued on `stream` (a hipStream_t passed as void*) and the call returns.
 *   - return value: YP_OK (0), a negative YP_ERR_* argument error, or a positive
 *     hipError_t from a failed launch.
 *   - workspace: caller-allocated device scratch, 256-byte aligned, at least the size the
 *     matching *_workspace_bytes function returns.  Calls that share a workspace must be
 *     ordered on one stream.
 *   - all arithmetic is IEEE binary32 without contraction; exp is the fixed sequence of
 *     floor/fma/mul/add operations documented in csrc/yp_math.h, so results are
 *     reproducible bit for bit.
 */
#ifndef YOLOPOST_H
#define YOLOPOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YP_MAX_HEADS 4
#define YP_MAX_ANCHORS 8
#define YP_MAX_SETS 8 /* independent batches one yp_detect_yolov3_multi call can take */

enum { YP_MODE_YOLOV3 = 0, YP_MODE_REFERENCE = 1 };

/* YpParams.flags.
 * YP_FLAG_DENSE_NMS (YP_MODE_YOLOV3 entries): expect (image, class) segments with thousands of
 * candidates -- low confidence thresholds, large inputs.  Segments above 768 candidates are then
 * finished by two extra kernels that spread one segment over many workgroups (pair tests over a
 * spatial index, then a fixed-point resolution of the keep set) instead of by their one
 * workgroup; results are bit-identical either way.  Without the flag nothing is launched for it.
 * Needs nms_thr in [0.05, 1] and boxes of positive size; segments that do not qualify fall back
 * to the one-workgroup path by themselves.
 *
 * YP_FLAG_RESCALE (yp_detect_yolov3, _multi, _noalloc): the kept boxes are mapped from the
 * network-input frame back to the original image, the step that follows write_results in the
 * YOLOv3 tutorial and the counterpart of the reference scaling its boxes to the camera frame
 * (detect_net.py:115-116, :133).  params->im_h / im_w give the original frame (one size for the
 * whole batch, as params.im_height / im_width are in the reference); the input size is grid x
 * stride of the first head.  With scale = min(in_w/im_w, in_h/im_h) and the letterbox pads
 * pad = (in - scale*im)/2, every corner becomes clamp((v - pad)/scale, 0, im).  NMS runs BEFORE
 * this, on the boxes in network-input pixels, as in the tutorial (in YP_MODE_REFERENCE the
 * scaling precedes NMS, as in the reference). */
enum { YP_FLAG_DENSE_NMS = 1, YP_FLAG_RESCALE = 2 };

enum {
    YP_OK = 0,
    YP_ERR_ARG = -1,        /* null pointer, non-positive size, value out of range */
    YP_ERR_WORKSPACE = -2,  /* workspace missing, misaligned or too small */
    YP_ERR_UNSUPPORTED = -3, /* shape beyond the compiled limits (heads, anchors, sizes) */
    YP_ERR_SCRATCH = -4      /* yp_detect_yolov3_noalloc: needs more than the static scratch */
};

/* One detection head.
 * YP_MODE_YOLOV3:    data is NCHW (B, A*(5+C), grid_h, grid_w); anchors[] holds
 *                    float32(anchor_px / stride) pairs (w,h); stride_x/stride_y are the
 *                    pixel strides (input_w / grid_w, input_h / grid_h).
 * YP_MODE_REFERENCE: data is (B, grid_h*grid_w*A*(5+C)) in memory order
 *                    (b,row,col,anchor,attr) (detect_net.py:153); anchors[] holds
 *                    params.YOLO_ANCHORS (w,h) in grid units (params.py:20-22); strides are
 *                    ignored. */
typedef struct YpHead {
    const float *data;
    int32_t grid_h;
    int32_t grid_w;
    int32_t num_anchors;
    int32_t reserved;
    float stride_x;
    float stride_y;
    float anchors[2 * YP_MAX_ANCHORS];
} YpHead;

/* Thresholds and sizes (params.py:9-10,15,27-28,31 in YP_MODE_REFERENCE). */
typedef struct YpParams {
    int32_t mode;         /* YP_MODE_* */
    int32_t batch;        /* B */
    int32_t num_classes;  /* C */
    int32_t max_out;      /* per-image cap (params.max_boxes).  YP_MODE_YOLOV3: <= 0 means
                             unlimited.  YP_MODE_REFERENCE entries pad their outputs to max_out
                             rows per image and require 1 <= max_out <= 2^20. */
    float conf_thr;       /* params.object_conf / `confidence` */
    float nms_thr;        /* params.nms_conf / `nms_conf` */
    float im_h;           /* YP_MODE_REFERENCE: params.im_height; YP_FLAG_RESCALE: frame height */
    float im_w;           /* YP_MODE_REFERENCE: params.im_width;  YP_FLAG_RESCALE: frame width */
    int32_t batch_offset; /* YP_MODE_YOLOV3: added to the image index written in column 0 */
    int32_t flags;        /* YP_FLAG_* bits; any other bit set is YP_ERR_ARG */
} YpParams;

int yp_version(void);
const char *yp_strerror(int code);

/* ---- decode only ------------------------------------------------------------------ */

/* Replaces DetectNet.predict_transform, detect_net.py:152-183.
 * d_pred (B, H*W*A*(5+C)) -> d_center (B,H,W,A,2), d_wh (B,H,W,A,2), d_obj (B,H,W,A,1),
 * d_cls (B,H,W,A,C).  anchors: HOST pointer to A (w,h) pairs. */
int yp_predict_transform_reference(const float *d_pred, int batch, int grid_h, int grid_w,
                                   int num_anchors, int num_classes, const float *anchors,
                                   float *d_center, float *d_wh, float *d_obj, float *d_cls,
                                   void *stream);

/* The head's producer with the decode in its epilogue: detect_net.py:67-68 (Dense(H*W*A*(5+C),
 * relu) on the flattened features) followed by detect_net.py:152-183 without storing the raw head
 * in between.  head[b, j] = relu(d_bias[j] + sum_f d_features[b, f] * d_weight[f, j]), summed in
 * the order f = 0, 1, ... in binary32 (a plain dot product per output: this entry shows the
 * epilogue hook csrc/yp_epilogue.h offers to whatever kernel produces the head; it is not a
 * tuned GEMM, and the model itself is outside this library).  d_features (B, F), d_weight
 * (F, H*W*A*(5+C)) row-major, d_bias (H*W*A*(5+C)); num_classes must be 5.  Outputs as
 * yp_predict_transform_reference, equal bit for bit to that entry applied to the stored head;
 * d_raw, if not NULL, receives the head (B, H*W*A*(5+C)) as well. */
int yp_dense_relu_predict_transform_reference(const float *d_features, const float *d_weight,
                                              const float *d_bias, int batch, int features, int grid_h,
                                              int grid_w, int num_anchors, int num_classes,
                                              const float *anchors, float *d_raw, float *d_center,
                                              float *d_wh, float *d_obj, float *d_cls, void *stream);

/* YOLOv3 predict_transform for 1..YP_MAX_HEADS heads in ONE launch (not in the reference;
 * see the mode note above).  Output rows (B, N, 5+C), N = sum_h gh*gw*A, heads
 * concatenated in array order, row index inside a head (row*gw+col)*A + a, row =
 * [cx, cy, w, h, objectness, class scores]. */
int yp_predict_transform_yolov3(const YpHead *heads, int n_heads, int batch, int num_classes,
                                float *d_rows, void *stream);

/* ---- whole path from raw heads (decode + filter + NMS fused) ------------------------- */

size_t yp_detect_yolov3_workspace_bytes(const YpHead *heads, int n_heads, int batch,
                                        int num_classes);

/* YOLOv3 path from raw heads: the composition predict_transform + write_results without
 * materialising the decoded rows.
 *   d_out_rows  (out_cap, 8) [image, x1, y1, x2, y2, objectness, class score, class id],
 *               ordered by image, class id ascending, objectness descending
 *   d_out_count [1] total detections D (rows beyond out_cap are not stored)
 *   d_stats     optional [4*B*C] int32, four planes of B*C per (image, class) segment:
 *               survivors of the filter, kept rows, IoU pairs evaluated (low 32 bits), and
 *               how many of those pairs had an IoU within 2^-20 (relative) of nms_thr -- the
 *               IoU-at-threshold ties, decided by the exact divide-and-compare; may be NULL */
int yp_detect_yolov3(const YpHead *heads, int n_heads, const YpParams *params,
                     float *d_out_rows, int64_t out_cap, int32_t *d_out_count,
                     int32_t *d_stats, void *d_workspace, size_t workspace_bytes,
                     void *stream);

/* Several independent batches ("sets") of identical geometry in ONE launch per stage.  The
 * reference calls the path once per batch (run_validation, train.py:345-348); a caller that
 * has n_sets batches ready (1..YP_MAX_SETS) hands them over together, which fills the GPU
 * with n_sets times the workgroups per launch and costs one third of the host calls.
 *   heads       n_sets * n_heads descriptors, set-major; every set repeats the geometry and
 *               anchors of set 0, only `data` differs
 *   params      shared by all sets; params->batch = images per set
 *   d_out_rows  (n_sets, out_cap, 8): set s writes its rows from d_out_rows + s*out_cap*8 (a
 *               row is 32 bytes, so every slab is 16-byte aligned when d_out_rows is); the image
 *               index in column 0 counts from params->batch_offset inside every set
 *   d_out_count [n_sets]
 *   d_stats     optional (n_sets, 4, B, C), planes as in yp_detect_yolov3
 *   workspace   yp_detect_yolov3_workspace_bytes(heads, n_heads, n_sets * batch, C)
 * Results are identical to n_sets calls of yp_detect_yolov3. */
int yp_detect_yolov3_multi(const YpHead *heads, int n_heads, int n_sets, const YpParams *params,
                           float *d_out_rows, int64_t out_cap, int32_t *d_out_count,
                           int32_t *d_stats, void *d_workspace, size_t workspace_bytes,
                           void *stream);

/* yp_detect_yolov3 for callers that cannot allocate device memory -- code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  It has no counterpart in
 * the reference.  Same inputs, outputs and results as yp_detect_yolov3, but the workspace is a
 * static device array inside the library (YP_NOALLOC_SCRATCH_BYTES = 16 MiB per GPU the
 * library runs on, held until the process ends); the call allocates and frees nothing.
 * Problems whose workspace (yp_detect_yolov3_workspace_bytes) exceeds that size return
 * YP_ERR_SCRATCH and launch nothing: 16 MiB covers e.g. 12 images at 416x416 or one image at
 * 1920x1280 with three anchors per head.  All calls on one GPU share that GPU's array: calls
 * that overlap on one GPU (different streams or threads) race; order them on one stream.
 * The array's device address is looked up on the first call on each device;
 * yp_noalloc_prepare() does that lookup for the current device ahead of time (e.g. before a
 * graph capture starts). */
#define YP_NOALLOC_SCRATCH_BYTES (16u << 20)
size_t yp_noalloc_scratch_bytes(void);
int yp_noalloc_prepare(void);
int yp_detect_yolov3_noalloc(const YpHead *heads, int n_heads, const YpParams *params,
                             float *d_out_rows, int64_t out_cap, int32_t *d_out_count,
                             int32_t *d_stats, void *stream);

size_t yp_detect_reference_workspace_bytes(int batch, int grid_h, int grid_w, int num_anchors);

/* Replaces the composition run_validation uses, train.py:347-348:
 * DetectNet.filter_boxes(DetectNet.predict_transform(pred)), detect_net.py:90-183.
 * Outputs are padded to max_out rows per image:
 *   d_out_boxes (B,max_out,4) [y1,x1,y2,x2] pixels, d_out_scores (B,max_out),
 *   d_out_classes (B,max_out) int32, d_out_index (B,max_out) int32 = position in the
 *   image's compacted survivor list (the reference's nms_indexes, detect_net.py:136,144),
 *   d_out_counts (B) rows used, d_surv_counts (B) survivors of the >= filter. */
int yp_detect_reference(const YpHead *head, const YpParams *params, float *d_out_boxes,
                        float *d_out_scores, int32_t *d_out_classes, int32_t *d_out_index,
                        int32_t *d_out_counts, int32_t *d_surv_counts, void *d_workspace,
                        size_t workspace_bytes, void *stream);

/* ---- filter + NMS on already decoded tensors ----------------------------------------- */

/* segments_per_image: num_classes for yp_write_results.  (With 1 it is an upper bound of what the
 * reference-mode entries need -- yp_detect_reference_workspace_bytes(batch, rows, 1, 1) is their
 * exact size, which leaves out the arrays only YP_MODE_YOLOV3 uses.) */
size_t yp_rows_workspace_bytes(int batch, int rows_per_image, int segments_per_image);

/* YOLOv3 write_results on decoded rows (B, N, 5+C) (not in the reference; mode note
 * above).  Outputs as yp_detect_yolov3.  d_rows is not modified. */
int yp_write_results(const float *d_rows, int rows_per_image, const YpParams *params,
                     float *d_out_rows, int64_t out_cap, int32_t *d_out_count,
                     int32_t *d_stats, void *d_workspace, size_t workspace_bytes,
                     void *stream);

/* Replaces DetectNet.filter_boxes, detect_net.py:90-146, on the 4-tuple returned by
 * predict_transform; N = H*W*A rows per image.  Outputs as yp_detect_reference. */
int yp_filter_boxes_reference(const float *d_center, const float *d_wh, const float *d_obj,
                              const float *d_cls, int rows_per_image, const YpParams *params,
                              float *d_out_boxes, float *d_out_scores, int32_t *d_out_classes,
                              int32_t *d_out_index, int32_t *d_out_counts,
                              int32_t *d_surv_counts, void *d_workspace,
                              size_t workspace_bytes, void *stream);

/* YOLOv3 bbox_iou (not in the reference): IoU, pixel-inclusive (+1) form, of the single
 * box d_box1[0..3] with M boxes whose corners are the first four floats of rows spaced
 * row_stride floats apart. */
int yp_bbox_iou(const float *d_box1, const float *d_boxes2, int m, int row_stride,
                float *d_out, void *stream);

/* ---- ground-truth grid codec (the label-side data format of the path) ---------------- */

/* Replaces create_true_box_grid, create_dataset.py:47-115, for a batch of images in one
 * launch.  d_objects (total, 5) float64 rows [cx, cy, w, h, class] in original pixels, the
 * objects of image b being rows d_object_offsets[b] .. d_object_offsets[b+1]-1 (B+1 int32
 * offsets on the device).  Writes every element of d_grid (B, H, W, A, 5) float32
 * [cx - col, cy - row, log(w / anchor_w), log(h / anchor_h), class] in grid units and of
 * d_mask (B, H, W, A, 2) float32 (1, 1) at occupied slots, zeros elsewhere.  Arithmetic is
 * binary64 with float32 anchors, as in the numpy code; when several objects of an image claim
 * one (row, col, anchor) slot the last one in list order wins, as in the reference's loop.
 * An object whose cell lies outside the grid is skipped (the reference raises IndexError or
 * wraps a negative index).  anchors: HOST pointer to A (w, h) pairs in grid units. */
int yp_encode_true_box_grid(const double *d_objects, const int32_t *d_object_offsets, int batch,
                            int grid_h, int grid_w, int num_anchors, const float *anchors,
                            int im_h, int im_w, float *d_grid, float *d_mask, void *stream);

size_t yp_true_box_grid_workspace_bytes(int batch, int grid_h, int grid_w, int num_anchors);

/* Replaces the grid decode of gather_grid_boxes, create_dataset.py:294-328, and the location
 * extraction of get_metrics, train.py:332-338, for a batch of grids (B, H, W, A, 5):
 *   d_out_index (out_cap, 4) int32 [image, row, col, anchor] of every slot with a non-zero
 *               entry, in ascending lexicographic order (np.unique order)
 *   d_out_boxes (out_cap, 4) float32 [cx, cy, w, h] in original pixels
 *   d_out_class (out_cap)    float32 class value stored in the slot
 *   d_out_count [1]          number of occupied slots (rows beyond out_cap are not stored) */
int yp_decode_true_box_grid(const float *d_grid, int batch, int grid_h, int grid_w,
                            int num_anchors, const float *anchors, int im_h, int im_w,
                            int32_t *d_out_index, float *d_out_boxes, float *d_out_class,
                            int64_t out_cap, int32_t *d_out_count, void *d_workspace,
                            size_t workspace_bytes, void *stream);

/* ---- validation statistics ----------------------------------------------------------- */

/* Kept detections against the ground truth of the same images: the comparison that
 * train.py:365-366 describes ("check iou of each nms box with each ground truth ..., if above
 * threshold compare classification") and leaves to the empty comp_nms_gt, train.py:279-280
 * (get_metrics, train.py:331-340, only prints the two location lists).
 *   d_det_rows (det_cap, 8) float32, 16-byte aligned: the detection list of yp_write_results /
 *       yp_detect_yolov3 [image, x1, y1, x2, y2, objectness, class score, class id], ordered by
 *       descending confidence within an image and class (as those calls emit it); the first
 *       min(*d_det_count, det_cap) rows are used
 *   d_gt_boxes (gt_cap, 4) [cx, cy, w, h], d_gt_class (gt_cap,), d_gt_index (gt_cap, 4) int32
 *       [image, row, col, anchor] ascending by image, *d_gt_count rows: the outputs of
 *       yp_decode_true_box_grid.  Boxes and detections must be in the same pixel frame.
 * Rule (PASCAL VOC): a detection's candidate is the ground-truth box of its image and class with
 * the highest IoU (the lowest row among equals); it is a true positive when that IoU >= iou_thr
 * (0 < iou_thr <= 1) and no earlier row of the list claimed the same box.  IoU is the plain
 * geometric one (no +1 term) in binary32: corners of the ground-truth box c -/+ size*0.5,
 * inter = max(min(x2) - max(x1), 0) * max(min(y2) - max(y1), 0), areas from the corners,
 * union = (area_det + area_gt) - inter, IoU = inter / union (0 when union <= 0).
 *   d_det_match [det_cap] int32: the ground-truth row a true positive claims, or YP_MATCH_NONE
 *       (no box of its class reaches the threshold, and none of another class), YP_MATCH_DUPLICATE
 *       (its candidate reached the threshold but belongs to an earlier detection),
 *       YP_MATCH_OTHER_CLASS (only a box of another class reaches the threshold)
 *   d_det_iou [det_cap] float32 or NULL: highest IoU with any ground-truth box of the image
 *   d_class_counts [YP_MATCH_PLANES * C] int32: per class id, plane 0 true positives, 1 false
 *       positives (all kinds), 2 misses (ground-truth boxes nobody claimed), 3 duplicates,
 *       4 other-class; class ids outside [0, C) are not counted
 * Deterministic (integer atomics only).  Two kernels and two memsets on the stream; no
 * allocation, no host synchronisation. */
#define YP_MATCH_PLANES 5
enum { YP_MATCH_NONE = -1, YP_MATCH_DUPLICATE = -2, YP_MATCH_OTHER_CLASS = -3 };
size_t yp_match_detections_workspace_bytes(int64_t gt_cap);
int yp_match_detections(const float *d_det_rows, const int32_t *d_det_count, int64_t det_cap,
                        const float *d_gt_boxes, const float *d_gt_class,
                        const int32_t *d_gt_index, const int32_t *d_gt_count, int64_t gt_cap,
                        int num_classes, float iou_thr, int32_t *d_det_match, float *d_det_iou,
                        int32_t *d_class_counts, void *d_workspace, size_t workspace_bytes,
                        void *stream);

/* ---- backward of the reference decode (for the training losses) ---------------------- */

/* Gradient of DetectNet.predict_transform, detect_net.py:152-183, with respect to its input,
 * as loss_custom / loss_keras need it (train.py:123-197, :200-260; the reference gets it from
 * tf.GradientTape, train.py:264-267).  d_grad_center (B,H,W,A,2), d_grad_wh (B,H,W,A,2),
 * d_grad_obj (B,H,W,A,1), d_grad_cls (B,H,W,A,C) are the upstream gradients of the four
 * outputs (any of them may be NULL = zero); d_grad_pred (B, H*W*A*(5+C)) receives the
 * gradient of the raw head.  binary32 throughout; num_classes <= 64. */
int yp_predict_transform_reference_backward(const float *d_pred, int batch, int grid_h,
                                            int grid_w, int num_anchors, int num_classes,
                                            const float *anchors, const float *d_grad_center,
                                            const float *d_grad_wh, const float *d_grad_obj,
                                            const float *d_grad_cls, float *d_grad_pred,
                                            void *stream);

size_t yp_loss_custom_reference_workspace_bytes(int batch, int grid_h, int grid_w,
                                                int num_anchors);

/* Replaces loss_custom, train.py:123-197 (and loss_keras, :200-270, the same arithmetic),
 * from the raw head: value and gradient in one pass.
 *   d_pred (B, H*W*A*(5+C)) raw head, d_true_grid (B,H,W,A,5) target grid
 *   weights: params.coord_loss_weight, obj_loss_weight, noobj_loss_weight (params.py:11-13);
 *   iou_thr: the 0.6 of train.py:167
 *   d_loss  [4] float32: total (= 0.5 * sum of the next three), confidence, classification,
 *           coordinates sums
 *   d_grad_pred (B, H*W*A*(5+C)) d total / d d_pred, or NULL for the value only
 * The detection indicator (IoU > iou_thr) is a constant for the gradient, as in TensorFlow. */
int yp_loss_custom_reference(const float *d_pred, const float *d_true_grid, int batch,
                             int grid_h, int grid_w, int num_anchors, int num_classes,
                             const float *anchors, float coord_weight, float obj_weight,
                             float noobj_weight, float iou_thr, float *d_loss,
                             float *d_grad_pred, void *d_workspace, size_t workspace_bytes,
                             void *stream);

/* ---- measurement -------------------------------------------------------------------- */

/* Per-kernel timing for benchmarks (no counterpart in the reference).  Between
 * yp_profile_start and yp_profile_stop every kernel of a yp_detect_*, yp_write_results or
 * yp_filter_boxes_reference call is launched with a start and a stop HIP event of its own
 * (hipExtLaunchKernelGGL), on the stream the call uses; at most max_calls calls are
 * recorded.  yp_profile_stop waits for the recorded events and returns the average
 * execution time in milliseconds per call of the front (decode + filter + compaction) kernel,
 * the NMS stage and the final gather kernel.  Not thread-safe; intended for a single
 * measuring thread. */
int yp_profile_start(int max_calls);
int yp_profile_stop(double *front_ms, double *nms_ms, double *gather_ms, int *calls);
/* The same with the kernels of the NMS stage apart: ms[5] = average milliseconds per recorded
 * call of the front kernel, the segment NMS kernel, the row gather, and the two kernels that
 * finish large segments (pair tests over the spatial index; resolution of the keep set) --
 * yp_profile_stop adds the last two to nms_ms. */
int yp_profile_stop_ex(double *ms, int *calls);

#ifdef __cplusplus
}
#endif
#endif /* YOLOPOST_H */
