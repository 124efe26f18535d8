/*
 * ypo.c -- CPU ORACLE for the YOLO detection post-processing path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product package may include, link,
 * load or call this file; it exists so that tests, the smoke check and the
 * bench's cpu_baseline leg have an independent, single-threaded restatement of
 * the algorithm to compare the HIP kernels against.
 *
 * Two semantic modes are restated:
 *
 *   Mode R ("reference"): /root/reference/detect_net.py as written.
 *     ypo_r_predict_transform   detect_net.py:152-183  (decode)
 *     ypo_r_filter_boxes        detect_net.py:90-146   (corners, score, >= filter,
 *                                                       boolean_mask order, image scale,
 *                                                       class-agnostic NMS, gather)
 *     ypo_tf_iou / ypo_tf_nms   restate tf.image.non_max_suppression, the call at
 *                               detect_net.py:136.  TensorFlow is a third-party
 *                               dependency that is NOT vendored in the reference and is
 *                               not installed here (README.md:11 implies TF 2.0.0).  The
 *                               arithmetic follows the published TF kernel
 *                               (non_max_suppression_op.cc): greedy by score, ties by
 *                               lower index, suppress when IoU > threshold, IoU without
 *                               "+1", corners normalised by min/max, zero-area -> IoU 0.
 *
 *   Mode N ("YOLOv3 tutorial"): the three-stride / per-class semantics named by the
 *     project brief (predict_transform / write_results / bbox_iou of the tutorial the
 *     reference credits at README.md:1).  Those functions are NOT present in
 *     /root/reference, so there is no file:line to cite; the semantics restated here are
 *     the tutorial's published algorithm.
 *
 * PARITY UNPINNED: the reference has no tests, golden vectors or fixtures, and it
 * cannot be imported here (TensorFlow absent).  This oracle is pinned only by
 * hand-derivable known answers (tests/test_oracle.py), by an independent numpy
 * restatement (oracle/np_oracle.py) and by the reference's own encode formula
 * (create_dataset.py:101-110) used as a round trip.  The one routine of the path that lives
 * in a third-party library -- tf.image.non_max_suppression, called at detect_net.py:136;
 * TensorFlow is neither vendored nor installed, README.md:11 implies version 2.0.0 -- is pinned
 * to the cases of TensorFlow's own unit test of the op (tests/tf_nms_vectors.py).
 *
 * exp flavours: flavour 0 uses libm expf.  Flavour 1 uses ypo_exp_spec, a fully
 * specified sequence of IEEE-754 binary32 operations (floor, fma, mul, add) that any
 * conforming implementation evaluates to identical bits; it lets whole-pipeline
 * results be compared bit for bit.  Its accuracy is checked against double exp in the
 * tests (<= 2 ulp), i.e. far inside the 1e-5 relative tolerance of the task.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC ypo.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float ypo_bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t ypo_f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* ------------------------------------------------------------------ exp / sigmoid */

float ypo_exp_spec(float x)
{
    if (x != x) return x;
    if (x > 88.72283935546875f) return INFINITY;
    if (x < -103.97207641601562f) return 0.0f;
    /* k = round-half-up(x / ln2), r = x - k ln2 with a two-term ln2 */
    float kf = floorf(fmaf(x, 1.44269504088896341f, 0.5f));
    float r = fmaf(kf, -0.693359375f, x);
    r = fmaf(kf, 2.12194440e-4f, r);
    /* degree-5 minimax tail (Cephes expf coefficients), Horner with fma */
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    p = fmaf(p, r2, r);
    p = p + 1.0f;
    /* scale by 2^k in two exact-power steps so every factor is a normal number */
    int k = (int)kf;
    int k1 = k >> 1;
    int k2 = k - k1;
    float s1 = ypo_bits2f((uint32_t)(k1 + 127) << 23);
    float s2 = ypo_bits2f((uint32_t)(k2 + 127) << 23);
    return (p * s1) * s2;
}

static inline float ypo_exp(float x, int flavour)
{
    return flavour ? ypo_exp_spec(x) : expf(x);
}

float ypo_sigmoid(float x, int flavour)
{
    return 1.0f / (1.0f + ypo_exp(-x, flavour));
}

/* --------------------------------------------------------------------- Mode R decode
 * detect_net.py:152-183.
 *   pred    (B, H*W*A*V)  row-major, memory order (b,row,col,anchor,attr)   :153
 *   grid    for flat cell k = row*W + col the offset pair is (k mod H, k div H) :155-162
 *           (equals (col,row) on the reference's square 13x13 grid)
 *   dims    channel 0 divided by H, channel 1 by W                           :163,168,177
 *   anchors (A,2) (w,h) in grid units, params.py:20-22                        :173-174
 */
void ypo_r_predict_transform(const float *pred, int B, int H, int W, int A, int C,
                             const float *anchors, int flavour,
                             float *center, float *wh, float *obj, float *cls)
{
    const int V = 5 + C;
    for (int b = 0; b < B; ++b)
        for (int r = 0; r < H; ++r)
            for (int c = 0; c < W; ++c)
                for (int a = 0; a < A; ++a) {
                    const long o = (((long)b * H + r) * W + c) * A + a;
                    const float *t = pred + o * V;
                    const int k = r * W + c;
                    const float gx = (float)(k % H);
                    const float gy = (float)(k / H);
                    center[o * 2 + 0] = (ypo_sigmoid(t[0], flavour) + gx) / (float)H;
                    center[o * 2 + 1] = (ypo_sigmoid(t[1], flavour) + gy) / (float)W;
                    obj[o] = ypo_sigmoid(t[4], flavour);
                    wh[o * 2 + 0] = (ypo_exp(t[2], flavour) * anchors[a * 2 + 0]) / (float)H;
                    wh[o * 2 + 1] = (ypo_exp(t[3], flavour) * anchors[a * 2 + 1]) / (float)W;
                    /* softmax over the class logits, detect_net.py:180 */
                    float m = t[5];
                    for (int j = 1; j < C; ++j)
                        if (t[5 + j] > m) m = t[5 + j];
                    float s = 0.0f;
                    for (int j = 0; j < C; ++j) {
                        float e = ypo_exp(t[5 + j] - m, flavour);
                        cls[o * C + j] = e;
                        s = s + e;
                    }
                    for (int j = 0; j < C; ++j) cls[o * C + j] = cls[o * C + j] / s;
                }
}

/* ------------------------------------------------------------------ TF-style IoU / NMS
 * boxes are [y1,x1,y2,x2]; restates tf.image.non_max_suppression (call site
 * detect_net.py:136; default score_threshold = -inf). */
float ypo_tf_iou(const float *bi, const float *bj)
{
    const float ymin_i = bi[0] < bi[2] ? bi[0] : bi[2];
    const float xmin_i = bi[1] < bi[3] ? bi[1] : bi[3];
    const float ymax_i = bi[0] > bi[2] ? bi[0] : bi[2];
    const float xmax_i = bi[1] > bi[3] ? bi[1] : bi[3];
    const float ymin_j = bj[0] < bj[2] ? bj[0] : bj[2];
    const float xmin_j = bj[1] < bj[3] ? bj[1] : bj[3];
    const float ymax_j = bj[0] > bj[2] ? bj[0] : bj[2];
    const float xmax_j = bj[1] > bj[3] ? bj[1] : bj[3];
    const float area_i = (ymax_i - ymin_i) * (xmax_i - xmin_i);
    const float area_j = (ymax_j - ymin_j) * (xmax_j - xmin_j);
    if (area_i <= 0.0f || area_j <= 0.0f) return 0.0f;
    const float iy0 = ymin_i > ymin_j ? ymin_i : ymin_j;
    const float ix0 = xmin_i > xmin_j ? xmin_i : xmin_j;
    const float iy1 = ymax_i < ymax_j ? ymax_i : ymax_j;
    const float ix1 = xmax_i < xmax_j ? xmax_i : xmax_j;
    float dh = iy1 - iy0;
    float dw = ix1 - ix0;
    if (!(dh > 0.0f)) dh = 0.0f;
    if (!(dw > 0.0f)) dw = 0.0f;
    const float inter = dh * dw;
    return inter / (area_i + area_j - inter);
}

typedef struct { float score; int idx; } ypo_cand;

static int ypo_cand_cmp(const void *pa, const void *pb)
{
    const ypo_cand *a = (const ypo_cand *)pa, *b = (const ypo_cand *)pb;
    if (a->score > b->score) return -1;
    if (a->score < b->score) return 1;
    return (a->idx > b->idx) - (a->idx < b->idx);
}

/* returns number selected; out_idx are positions in the input list, score-descending */
int ypo_tf_nms(const float *boxes, const float *scores, int K, int max_out, float iou_thr,
               int *out_idx)
{
    if (K <= 0 || max_out <= 0) return 0;
    ypo_cand *cand = (ypo_cand *)malloc(sizeof(ypo_cand) * (size_t)K);
    int nc = 0;
    for (int i = 0; i < K; ++i)
        if (scores[i] > -INFINITY) { cand[nc].score = scores[i]; cand[nc].idx = i; ++nc; }
    qsort(cand, (size_t)nc, sizeof(ypo_cand), ypo_cand_cmp);
    int ns = 0;
    for (int q = 0; q < nc && ns < max_out; ++q) {
        const int i = cand[q].idx;
        int keep = 1;
        for (int s = ns - 1; s >= 0; --s)
            if (ypo_tf_iou(boxes + 4 * (long)i, boxes + 4 * (long)out_idx[s]) > iou_thr) {
                keep = 0;
                break;
            }
        if (keep) out_idx[ns++] = i;
    }
    free(cand);
    return ns;
}

/* --------------------------------------------------------------- Mode R filter_boxes
 * detect_net.py:90-146.  Outputs are padded per image to max_boxes rows;
 * out_counts[b] = rows used, surv_counts[b] = survivors of the >= filter (K).
 * out_index = positions in the image's compacted survivor list (what the reference
 * returns as nms_indexes, detect_net.py:136,144). */
void ypo_r_filter_boxes(const float *center, const float *wh, const float *obj,
                        const float *cls, int B, int N, int C, float conf_thr,
                        float nms_thr, int max_boxes, float im_h, float im_w,
                        float *out_boxes, float *out_scores, int64_t *out_classes,
                        int32_t *out_index, int32_t *out_counts, int32_t *surv_counts)
{
    float *sb = (float *)malloc(sizeof(float) * 4 * (size_t)(N > 0 ? N : 1));
    float *ss = (float *)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1));
    int *sc = (int *)malloc(sizeof(int) * (size_t)(N > 0 ? N : 1));
    int *sel = (int *)malloc(sizeof(int) * (size_t)(max_boxes > 0 ? max_boxes : 1));
    for (int b = 0; b < B; ++b) {
        int K = 0;
        for (int n = 0; n < N; ++n) {
            const long o = (long)b * N + n;
            const float cx = center[o * 2 + 0], cy = center[o * 2 + 1];
            const float hw = wh[o * 2 + 0] / 2.0f, hh = wh[o * 2 + 1] / 2.0f; /* :95-96 */
            const float minx = cx - hw, miny = cy - hh;
            const float maxx = cx + hw, maxy = cy + hh;
            /* score = obj * class prob, argmax (first max wins), max           :107-109 */
            float best = obj[o] * cls[o * C + 0];
            int bi = 0;
            for (int j = 1; j < C; ++j) {
                const float s = obj[o] * cls[o * C + j];
                if (s > best) { best = s; bi = j; }
            }
            if (best >= conf_thr) {                                           /* :111 */
                sb[K * 4 + 0] = miny * im_h;                                   /* :97-102,133 */
                sb[K * 4 + 1] = minx * im_w;
                sb[K * 4 + 2] = maxy * im_h;
                sb[K * 4 + 3] = maxx * im_w;
                ss[K] = best;
                sc[K] = bi;
                ++K;
            }
        }
        surv_counts[b] = K;
        const int M = ypo_tf_nms(sb, ss, K, max_boxes, nms_thr, sel);         /* :136 */
        out_counts[b] = M;
        for (int m = 0; m < M; ++m) {                                          /* :137-139 */
            const long q = (long)b * max_boxes + m;
            memcpy(out_boxes + q * 4, sb + 4 * (long)sel[m], 16);
            out_scores[q] = ss[sel[m]];
            out_classes[q] = sc[sel[m]];
            out_index[q] = sel[m];
        }
    }
    free(sb); free(ss); free(sc); free(sel);
}

/* --------------------------------------------------------------------- Mode N decode
 * One detection head, NCHW (B, A*V, gh, gw).  Output rows (B, gh*gw*A, V) with row index
 * (row*gw + col)*A + a and row = [cx, cy, w, h, obj, cls_0..cls_{C-1}] in input pixels.
 * anchors_scaled = float32(anchor_px / stride) (the tutorial divides in double and then
 * converts the list to a float tensor). */
void ypo_n_predict_transform(const float *head, int B, int A, int C, int gh, int gw,
                             float stride_x, float stride_y, const float *anchors_scaled,
                             int flavour, float *rows)
{
    const int V = 5 + C;
    const long cells = (long)gh * gw;
    for (int b = 0; b < B; ++b)
        for (int r = 0; r < gh; ++r)
            for (int c = 0; c < gw; ++c)
                for (int a = 0; a < A; ++a) {
                    const long cell = (long)r * gw + c;
                    const float *t = head + ((long)b * A + a) * V * cells + cell;
                    float *o = rows + (((long)b * cells + cell) * A + a) * V;
                    o[0] = (ypo_sigmoid(t[0 * cells], flavour) + (float)c) * stride_x;
                    o[1] = (ypo_sigmoid(t[1 * cells], flavour) + (float)r) * stride_y;
                    o[2] = (ypo_exp(t[2 * cells], flavour) * anchors_scaled[a * 2 + 0]) * stride_x;
                    o[3] = (ypo_exp(t[3 * cells], flavour) * anchors_scaled[a * 2 + 1]) * stride_y;
                    o[4] = ypo_sigmoid(t[4 * cells], flavour);
                    for (int j = 0; j < C; ++j)
                        o[5 + j] = ypo_sigmoid(t[(5 + j) * cells], flavour);
                }
}

/* tutorial bbox_iou: corners x1,y1,x2,y2, pixel-inclusive (+1) */
float ypo_n_iou(const float *b1, const float *b2)
{
    const float ix1 = b1[0] > b2[0] ? b1[0] : b2[0];
    const float iy1 = b1[1] > b2[1] ? b1[1] : b2[1];
    const float ix2 = b1[2] < b2[2] ? b1[2] : b2[2];
    const float iy2 = b1[3] < b2[3] ? b1[3] : b2[3];
    float dw = ix2 - ix1 + 1.0f;
    float dh = iy2 - iy1 + 1.0f;
    if (!(dw > 0.0f)) dw = 0.0f;
    if (!(dh > 0.0f)) dh = 0.0f;
    const float inter = dw * dh;
    const float a1 = (b1[2] - b1[0] + 1.0f) * (b1[3] - b1[1] + 1.0f);
    const float a2 = (b2[2] - b2[0] + 1.0f) * (b2[3] - b2[1] + 1.0f);
    return inter / (a1 + a2 - inter);
}

void ypo_n_bbox_iou(const float *box1, const float *boxes2, int M, int row_stride, float *out)
{
    for (int m = 0; m < M; ++m) out[m] = ypo_n_iou(box1, boxes2 + (long)m * row_stride);
}

typedef struct { float key; int idx; } ypo_kc;
static int ypo_kc_cmp(const void *pa, const void *pb)
{
    const ypo_kc *a = (const ypo_kc *)pa, *b = (const ypo_kc *)pb;
    if (a->key > b->key) return -1;
    if (a->key < b->key) return 1;
    return (a->idx > b->idx) - (a->idx < b->idx);
}

/* tutorial write_results on decoded rows (B, N, 5+C).
 *  - survivors: objectness > confidence (strict) and objectness != 0
 *  - corners x1=cx-w/2 ... ; class score = max, class id = argmax (first max wins)
 *  - per image, per class id ascending: sort by objectness descending (ties: lower
 *    anchor index first), greedy: a later row is dropped unless IoU < nms_conf
 *  - output rows [batch, x1,y1,x2,y2, obj, cls_score, cls_id], at most cap rows stored
 * returns the total number of detections D (may exceed cap).  Optional statistics:
 * surv_counts[b] = survivors per image, pairs[0] += IoU evaluations. */
long ypo_n_write_results(const float *rows, int B, int N, int C, float confidence,
                         float nms_conf, float *out, long cap, int32_t *surv_counts,
                         int64_t *pairs)
{
    const int V = 5 + C;
    float *sb = (float *)malloc(sizeof(float) * 4 * (size_t)(N > 0 ? N : 1));
    float *so = (float *)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1));
    float *sc = (float *)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1));
    int *sid = (int *)malloc(sizeof(int) * (size_t)(N > 0 ? N : 1));
    ypo_kc *ord = (ypo_kc *)malloc(sizeof(ypo_kc) * (size_t)(N > 0 ? N : 1));
    char *dead = (char *)malloc((size_t)(N > 0 ? N : 1));
    long D = 0;
    int64_t npairs = 0;
    for (int b = 0; b < B; ++b) {
        int K = 0;
        for (int n = 0; n < N; ++n) {
            const float *p = rows + ((long)b * N + n) * V;
            const float o = p[4];
            if (!(o > confidence) || o == 0.0f) continue;
            const float hw = p[2] / 2.0f, hh = p[3] / 2.0f;
            sb[K * 4 + 0] = p[0] - hw;
            sb[K * 4 + 1] = p[1] - hh;
            sb[K * 4 + 2] = p[0] + hw;
            sb[K * 4 + 3] = p[1] + hh;
            float best = p[5];
            int bi = 0;
            for (int j = 1; j < C; ++j)
                if (p[5 + j] > best) { best = p[5 + j]; bi = j; }
            so[K] = o + 0.0f;
            sc[K] = best;
            sid[K] = bi;
            ++K;
        }
        if (surv_counts) surv_counts[b] = K;
        for (int c = 0; c < C; ++c) {
            int m = 0;
            for (int k = 0; k < K; ++k)
                if (sid[k] == c) { ord[m].key = so[k]; ord[m].idx = k; ++m; }
            if (m == 0) continue;
            qsort(ord, (size_t)m, sizeof(ypo_kc), ypo_kc_cmp);
            memset(dead, 0, (size_t)m);
            for (int i = 0; i < m; ++i) {
                if (dead[i]) continue;
                const int ki = ord[i].idx;
                if (D < cap) {
                    float *q = out + D * 8;
                    q[0] = (float)b;
                    q[1] = sb[ki * 4 + 0]; q[2] = sb[ki * 4 + 1];
                    q[3] = sb[ki * 4 + 2]; q[4] = sb[ki * 4 + 3];
                    q[5] = so[ki]; q[6] = sc[ki]; q[7] = (float)c;
                }
                ++D;
                for (int j = i + 1; j < m; ++j) {
                    if (dead[j]) continue;
                    ++npairs;
                    const float iou = ypo_n_iou(sb + 4 * (long)ki, sb + 4 * (long)ord[j].idx);
                    if (!(iou < nms_conf)) dead[j] = 1;
                }
            }
        }
    }
    if (pairs) pairs[0] += npairs;
    free(sb); free(so); free(sc); free(sid); free(ord); free(dead);
    return D;
}

/* Whole Mode-N path from raw heads: decode every head into one (B, N, V) row tensor in
 * head order, then write_results.  rows_ws must hold B*N*V floats.  Used as the CPU
 * baseline leg of the bench and by parity tests. */
long ypo_n_detect(const float *const *heads, int n_heads, const int *gh, const int *gw,
                  const int *na, const float *stride_x, const float *stride_y,
                  const float *const *anchors_scaled, int B, int C, float confidence,
                  float nms_conf, int flavour, float *rows_ws, float *out, long cap,
                  int32_t *surv_counts, int64_t *pairs)
{
    const int V = 5 + C;
    long N = 0;
    for (int h = 0; h < n_heads; ++h) N += (long)gh[h] * gw[h] * na[h];
    long off = 0;
    for (int h = 0; h < n_heads; ++h) {
        const long nh = (long)gh[h] * gw[h] * na[h];
        float *tmp = (float *)malloc(sizeof(float) * (size_t)B * nh * V);
        ypo_n_predict_transform(heads[h], B, na[h], C, gh[h], gw[h], stride_x[h], stride_y[h],
                                anchors_scaled[h], flavour, tmp);
        for (int b = 0; b < B; ++b)
            memcpy(rows_ws + ((long)b * N + off) * V, tmp + (long)b * nh * V,
                   sizeof(float) * (size_t)nh * V);
        free(tmp);
        off += nh;
    }
    return ypo_n_write_results(rows_ws, B, (int)N, C, confidence, nms_conf, out, cap,
                               surv_counts, pairs);
}

/* Whole Mode-R path from the raw head (B, H*W*A*V). */
void ypo_r_detect(const float *pred, int B, int H, int W, int A, int C, const float *anchors,
                  float conf_thr, float nms_thr, int max_boxes, float im_h, float im_w,
                  int flavour, float *out_boxes, float *out_scores, int64_t *out_classes,
                  int32_t *out_index, int32_t *out_counts, int32_t *surv_counts)
{
    const long n = (long)B * H * W * A;
    float *center = (float *)malloc(sizeof(float) * 2 * (size_t)n);
    float *wh = (float *)malloc(sizeof(float) * 2 * (size_t)n);
    float *obj = (float *)malloc(sizeof(float) * (size_t)n);
    float *cls = (float *)malloc(sizeof(float) * (size_t)n * C);
    ypo_r_predict_transform(pred, B, H, W, A, C, anchors, flavour, center, wh, obj, cls);
    ypo_r_filter_boxes(center, wh, obj, cls, B, H * W * A, C, conf_thr, nms_thr, max_boxes,
                       im_h, im_w, out_boxes, out_scores, out_classes, out_index, out_counts,
                       surv_counts);
    free(center); free(wh); free(obj); free(cls);
}
