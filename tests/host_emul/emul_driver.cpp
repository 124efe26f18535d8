// Replays one recorded call of yp_detect_yolov3_multi through the host emulation of the kernels
// (built with AddressSanitizer + UndefinedBehaviorSanitizer by the Makefile in this directory)
// and compares the output with the expected rows stored in the case file.  Every buffer is a heap
// block of exactly the size the ABI documents, so a kernel that reads or writes one element too far
// is reported by the sanitizer.  Case file (little endian), written by tests/test_y_emulation.py:
//   int32  magic 0x31455059, n_sets, batch, classes, n_heads, flags, cap, batch_offset
//   float  conf, nms, im_h, im_w
//   per head: int32 grid_h, grid_w, anchors; float stride_x, stride_y, anchors[16]
//   per set, per head: float data[batch * anchors * (5 + classes) * grid_h * grid_w]
//   per set: int32 count; float rows[min(count, cap) * 8]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "yolopost.h"

extern "C" void emul_stats(long *out, int reset);

static bool read_exact(FILE *f, void *p, size_t n) { return n == 0 || std::fread(p, 1, n, f) == n; }

// A recorded yp_match_detections call (magic 0x32455059):
//   int32 magic, n_det, n_gt, classes, det_cap, gt_cap, 0, 0;  float thr, 0, 0, 0
//   float rows[n_det * 8]; int32 gt_index[n_gt * 4]; float gt_boxes[n_gt * 4]; float gt_class[n_gt]
//   int32 det_match[n_det]; float det_iou[n_det]; int32 counts[5 * classes]
static int replay_match(FILE *f, const int32_t *h, float thr)
{
    const int D = h[1], G = h[2], C = h[3], det_cap = h[4], gt_cap = h[5];
    float *rows = (float *)std::aligned_alloc(16, (size_t)det_cap * 32);
    int32_t *gi = (int32_t *)std::malloc((size_t)gt_cap * 16);
    float *gb = (float *)std::aligned_alloc(16, (size_t)gt_cap * 16);
    float *gc = (float *)std::malloc((size_t)gt_cap * 4);
    for (int i = 0; i < det_cap * 8; ++i) rows[i] = 7.0f;       // rows beyond the count look like detections
    std::memset(gi, 0, (size_t)gt_cap * 16);
    for (int i = 0; i < gt_cap * 4; ++i) gb[i] = 50.0f;
    for (int i = 0; i < gt_cap; ++i) gc[i] = 1.0f;
    if (!read_exact(f, rows, (size_t)D * 32) || !read_exact(f, gi, (size_t)G * 16) || !read_exact(f, gb, (size_t)G * 16) ||
        !read_exact(f, gc, (size_t)G * 4))
        return 2;
    std::vector<int32_t> want_m(D), want_c((size_t)5 * C);
    std::vector<float> want_iou(D);
    if (!read_exact(f, want_m.data(), (size_t)D * 4) || !read_exact(f, want_iou.data(), (size_t)D * 4) ||
        !read_exact(f, want_c.data(), want_c.size() * 4))
        return 2;
    int32_t nd = D, ng = G;
    int32_t *match = (int32_t *)std::malloc((size_t)det_cap * 4);
    float *iou = (float *)std::malloc((size_t)det_cap * 4);
    int32_t *counts = (int32_t *)std::malloc((size_t)5 * C * 4);
    const size_t need = yp_match_detections_workspace_bytes(gt_cap);
    void *ws = std::aligned_alloc(256, need);
    std::memset(ws, 0xA5, need);
    std::memset(match, 0xA5, (size_t)det_cap * 4);
    const int rc = yp_match_detections(rows, &nd, det_cap, gb, gc, gi, &ng, gt_cap, C, thr, match, iou, counts, ws, need,
                                       nullptr);
    if (rc != 0) { std::printf("yp_match_detections returned %d\n", rc); return 1; }
    int bad = 0;
    if (std::memcmp(match, want_m.data(), (size_t)D * 4) != 0) { std::printf("det_match differs\n"); ++bad; }
    if (std::memcmp(iou, want_iou.data(), (size_t)D * 4) != 0) { std::printf("det_iou differs\n"); ++bad; }
    if (std::memcmp(counts, want_c.data(), want_c.size() * 4) != 0) { std::printf("class counts differ\n"); ++bad; }
    for (int i = D; i < det_cap; ++i)
        if (match[i] != (int32_t)0xA5A5A5A5) { std::printf("det_match written beyond the count\n"); ++bad; break; }
    std::printf("%s: matching, %d detections, %d ground-truth boxes\n", bad ? "MISMATCH" : "ok", D, G);
    return bad ? 1 : 0;
}

// A recorded yp_detect_reference call (magic 0x33455059):
//   int32 magic, batch, grid_h, grid_w, anchors, classes, max_out, 0;  float conf, nms, im_h, im_w
//   float anchors[2 * anchors]; float pred[batch * grid_h * grid_w * anchors * (5 + classes)]
//   int32 counts[batch], surv[batch]; then per image: float boxes[n*4], float scores[n], int32 classes[n], int32 index[n]
static int replay_reference(FILE *f, const int32_t *h, const float *t)
{
    const int B = h[1], H = h[2], W = h[3], A = h[4], C = h[5], mb = h[6];
    YpHead head;
    std::memset(&head, 0, sizeof head);
    head.grid_h = H; head.grid_w = W; head.num_anchors = A;
    if (!read_exact(f, head.anchors, (size_t)2 * A * 4)) return 2;
    const size_t n_pred = (size_t)B * H * W * A * (5 + C);
    float *pred = (float *)std::aligned_alloc(16, (n_pred * 4 + 15) / 16 * 16);
    if (!read_exact(f, pred, n_pred * 4)) return 2;
    head.data = pred;
    YpParams p;
    std::memset(&p, 0, sizeof p);
    p.mode = YP_MODE_REFERENCE; p.batch = B; p.num_classes = C; p.max_out = mb;
    p.conf_thr = t[0]; p.nms_thr = t[1]; p.im_h = t[2]; p.im_w = t[3];
    const size_t need = yp_detect_reference_workspace_bytes(B, H, W, A);
    void *ws = std::aligned_alloc(256, need);
    std::memset(ws, 0xA5, need);
    float *boxes = (float *)std::aligned_alloc(16, (size_t)B * mb * 16);
    float *scores = (float *)std::malloc((size_t)B * mb * 4);
    int32_t *classes = (int32_t *)std::malloc((size_t)B * mb * 4), *index = (int32_t *)std::malloc((size_t)B * mb * 4);
    int32_t *counts = (int32_t *)std::malloc((size_t)B * 4), *surv = (int32_t *)std::malloc((size_t)B * 4);
    const int rc = yp_detect_reference(&head, &p, boxes, scores, classes, index, counts, surv, ws, need, nullptr);
    if (rc != 0) { std::printf("yp_detect_reference returned %d\n", rc); return 1; }
    std::vector<int32_t> want_counts(B), want_surv(B);
    if (!read_exact(f, want_counts.data(), (size_t)B * 4) || !read_exact(f, want_surv.data(), (size_t)B * 4)) return 2;
    int bad = 0;
    if (std::memcmp(counts, want_counts.data(), (size_t)B * 4) != 0) { std::printf("counts differ\n"); ++bad; }
    if (std::memcmp(surv, want_surv.data(), (size_t)B * 4) != 0) { std::printf("survivor counts differ\n"); ++bad; }
    for (int b = 0; b < B && !bad; ++b) {
        const int n = want_counts[b];
        std::vector<float> wb((size_t)n * 4), wsc(n);
        std::vector<int32_t> wc(n), wi(n);
        if (!read_exact(f, wb.data(), wb.size() * 4) || !read_exact(f, wsc.data(), (size_t)n * 4) ||
            !read_exact(f, wc.data(), (size_t)n * 4) || !read_exact(f, wi.data(), (size_t)n * 4))
            return 2;
        if (std::memcmp(boxes + (size_t)b * mb * 4, wb.data(), wb.size() * 4) != 0 ||
            std::memcmp(scores + (size_t)b * mb, wsc.data(), (size_t)n * 4) != 0 ||
            std::memcmp(classes + (size_t)b * mb, wc.data(), (size_t)n * 4) != 0 ||
            std::memcmp(index + (size_t)b * mb, wi.data(), (size_t)n * 4) != 0) {
            std::printf("image %d differs\n", b);
            ++bad;
        }
    }
    long st[5];
    emul_stats(st, 0);
    std::printf("%s: reference mode, %d images, %ld divergent groups, %ld shuffles from inactive lanes\n",
                bad ? "MISMATCH" : "ok", B, st[1], st[2]);
    return bad || st[1] || st[2] ? 1 : 0;
}

int main(int argc, char **argv)
{
    if (argc != 2) return 2;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t h[8];
    float t[4];
    if (!read_exact(f, h, sizeof h) || !read_exact(f, t, sizeof t)) return 2;
    if (h[0] == 0x32455059) return replay_match(f, h, t[0]);
    if (h[0] == 0x33455059) return replay_reference(f, h, t);
    if (h[0] != 0x31455059) return 2;
    const int n_sets = h[1], B = h[2], C = h[3], n_heads = h[4], flags = h[5], cap = h[6];
    std::vector<YpHead> heads((size_t)n_sets * n_heads);
    std::vector<size_t> elems(n_heads);
    for (int k = 0; k < n_heads; ++k) {
        int32_t g[3];
        float s[2 + 2 * YP_MAX_ANCHORS];
        if (!read_exact(f, g, sizeof g) || !read_exact(f, s, sizeof s)) return 2;
        for (int set = 0; set < n_sets; ++set) {
            YpHead &d = heads[(size_t)set * n_heads + k];
            std::memset(&d, 0, sizeof d);
            d.grid_h = g[0]; d.grid_w = g[1]; d.num_anchors = g[2];
            d.stride_x = s[0]; d.stride_y = s[1];
            std::memcpy(d.anchors, s + 2, sizeof d.anchors);
        }
        elems[k] = (size_t)B * g[2] * (5 + C) * g[0] * g[1];
    }
    for (int set = 0; set < n_sets; ++set)
        for (int k = 0; k < n_heads; ++k) {
            float *data = (float *)std::aligned_alloc(16, (elems[k] * 4 + 15) / 16 * 16);
            if (!read_exact(f, data, elems[k] * 4)) return 2;
            heads[(size_t)set * n_heads + k].data = data;
        }
    YpParams p;
    std::memset(&p, 0, sizeof p);
    p.mode = YP_MODE_YOLOV3; p.batch = B; p.num_classes = C; p.max_out = 0;
    p.conf_thr = t[0]; p.nms_thr = t[1]; p.im_h = t[2]; p.im_w = t[3];
    p.batch_offset = h[7]; p.flags = flags;
    const size_t need = yp_detect_yolov3_workspace_bytes(heads.data(), n_heads, n_sets * B, C);
    if (need == 0) { std::printf("unsupported geometry\n"); return 1; }
    void *ws = std::aligned_alloc(256, need);
    float *out = (float *)std::aligned_alloc(16, (size_t)n_sets * cap * 32);
    int32_t *count = (int32_t *)std::malloc((size_t)n_sets * 4);
    int32_t *stats = (int32_t *)std::malloc((size_t)n_sets * 4 * B * C * 4);
    std::memset(ws, 0xA5, need);
    std::memset(out, 0xA5, (size_t)n_sets * cap * 32);
    const int rc = yp_detect_yolov3_multi(heads.data(), n_heads, n_sets, &p, out, cap, count, stats, ws, need, nullptr);
    if (rc != 0) { std::printf("yp_detect_yolov3_multi returned %d\n", rc); return 1; }
    int bad = 0;
    for (int set = 0; set < n_sets; ++set) {
        int32_t want;
        if (!read_exact(f, &want, 4)) return 2;
        const int n = want < cap ? want : cap;
        std::vector<float> rows((size_t)n * 8);
        if (!read_exact(f, rows.data(), rows.size() * 4)) return 2;
        if (count[set] != want) { std::printf("set %d: count %d, expected %d\n", set, count[set], want); ++bad; continue; }
        if (n > 0 && std::memcmp(rows.data(), out + (size_t)set * cap * 8, rows.size() * 4) != 0) {
            std::printf("set %d: rows differ\n", set);
            ++bad;
        }
        long kept = 0;
        for (int i = 0; i < B * C; ++i) kept += stats[((size_t)set * 4 + 1) * B * C + i];
        if (kept != want) { std::printf("set %d: statistics count %ld kept rows, expected %d\n", set, kept, want); ++bad; }
    }
    long st[5];
    emul_stats(st, 0);
    std::printf("%s: %d sets, %ld collectives, %ld divergent groups, %ld shuffles from inactive lanes, %ld workgroups\n",
                bad ? "MISMATCH" : "ok", n_sets, st[0], st[1], st[2], st[4]);
    return bad || st[1] || st[2] ? 1 : 0;
}
