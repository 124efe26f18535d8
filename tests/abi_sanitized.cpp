// Host-side exercise of the C ABI's argument validation and workspace layout code, meant to be
// built with the host sanitizers (AddressSanitizer + UndefinedBehaviorSanitizer) and run on a
// machine WITHOUT a GPU: every call either is rejected before anything is launched or fails at
// the launch with a HIP error -- no kernel ever runs.  Prints "ok <n>" when every check held.
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "yolopost.h"

static int g_fail = 0, g_checks = 0;
#define CHECK(cond)                                                        \
    do {                                                                   \
        ++g_checks;                                                        \
        if (!(cond)) { ++g_fail; std::printf("FAILED %s:%d %s\n", __FILE__, __LINE__, #cond); } \
    } while (0)

static YpHead head(int g, int anchors, float stride)
{
    YpHead h;
    std::memset(&h, 0, sizeof h);
    h.data = reinterpret_cast<const float *>(0x1000);
    h.grid_h = h.grid_w = g;
    h.num_anchors = anchors;
    h.stride_x = h.stride_y = stride;
    for (int i = 0; i < 2 * YP_MAX_ANCHORS; ++i) h.anchors[i] = 1.0f + i;
    return h;
}

int main()
{
    void *fake = reinterpret_cast<void *>(0x1000);
    float *ff = static_cast<float *>(fake);
    int32_t *fi = static_cast<int32_t *>(fake);
    YpParams p;
    std::memset(&p, 0, sizeof p);
    p.mode = YP_MODE_YOLOV3;
    p.batch = 2;
    p.num_classes = 5;
    p.conf_thr = 0.5f;
    p.nms_thr = 0.4f;

    CHECK(yp_version() >= 200);
    CHECK(std::strcmp(yp_strerror(YP_OK), "ok") == 0);
    for (int c = -6; c <= 2; ++c) CHECK(yp_strerror(c) != nullptr);

    // workspace sizes over a sweep of geometries: never zero for valid shapes, monotone in the batch
    YpHead hs[YP_MAX_SETS * 3];
    for (int s = 0; s < YP_MAX_SETS; ++s) {
        hs[3 * s + 0] = head(13, 3, 32.f);
        hs[3 * s + 1] = head(26, 3, 16.f);
        hs[3 * s + 2] = head(52, 3, 8.f);
    }
    size_t prev = 0;
    for (int b = 1; b <= 512; b *= 2) {
        const size_t w = yp_detect_yolov3_workspace_bytes(hs, 3, b, 5);
        CHECK(w > prev && w % 256 == 0);
        prev = w;
    }
    CHECK(yp_detect_yolov3_workspace_bytes(hs, 0, 1, 5) == 0);
    CHECK(yp_detect_yolov3_workspace_bytes(hs, 5, 1, 5) == 0);
    CHECK(yp_detect_yolov3_workspace_bytes(nullptr, 3, 1, 5) == 0);
    CHECK(yp_detect_yolov3_workspace_bytes(hs, 3, 1 << 30, 5) == 0);     // slot count overflows
    CHECK(yp_rows_workspace_bytes(16, 845, 1) > 0 && yp_rows_workspace_bytes(0, 845, 1) == 0);
    CHECK(yp_rows_workspace_bytes(1 << 20, 1 << 27, 1) == 0);
    CHECK(yp_detect_reference_workspace_bytes(16, 13, 13, 5) > 0 &&
          yp_detect_reference_workspace_bytes(16, 13, 13, 5) < yp_rows_workspace_bytes(16, 845, 1));
    CHECK(yp_true_box_grid_workspace_bytes(16, 13, 13, 5) > 0);
    CHECK(yp_loss_custom_reference_workspace_bytes(16, 13, 13, 5) > 0);

    // argument errors, each rejected before a launch
    const size_t big = (size_t)1 << 34;
    CHECK(yp_detect_yolov3(nullptr, 3, &p, ff, 10, fi, nullptr, fake, big, nullptr) == YP_ERR_ARG);
    CHECK(yp_detect_yolov3(hs, 3, nullptr, ff, 10, fi, nullptr, fake, big, nullptr) == YP_ERR_ARG);
    CHECK(yp_detect_yolov3(hs, 3, &p, nullptr, 10, fi, nullptr, fake, big, nullptr) == YP_ERR_ARG);
    CHECK(yp_detect_yolov3(hs, 3, &p, ff, -1, fi, nullptr, fake, big, nullptr) == YP_ERR_ARG);
    CHECK(yp_detect_yolov3(hs, 3, &p, ff + 1, 10, fi, nullptr, fake, big, nullptr) == YP_ERR_ARG);
    CHECK(yp_detect_yolov3(hs, 3, &p, ff, 10, fi, nullptr, fake, 16, nullptr) == YP_ERR_WORKSPACE);
    CHECK(yp_detect_yolov3(hs, 3, &p, ff, 10, fi, nullptr, nullptr, big, nullptr) == YP_ERR_WORKSPACE);
    CHECK(yp_detect_yolov3(hs, 3, &p, ff, 10, fi, nullptr, reinterpret_cast<void *>(0x1010), big, nullptr) ==
          YP_ERR_WORKSPACE);
    {
        YpParams q = p;
        q.flags = 4;                                   // unknown bit
        CHECK(yp_detect_yolov3(hs, 3, &q, ff, 10, fi, nullptr, fake, big, nullptr) == YP_ERR_ARG);
        q.flags = YP_FLAG_RESCALE;                     // needs a frame
        q.im_h = 0.0f; q.im_w = 1920.0f;
        CHECK(yp_detect_yolov3(hs, 3, &q, ff, 10, fi, nullptr, fake, big, nullptr) == YP_ERR_ARG);
        q.im_h = 1280.0f; q.im_w = __builtin_nanf("");
        CHECK(yp_detect_yolov3(hs, 3, &q, ff, 10, fi, nullptr, fake, big, nullptr) == YP_ERR_ARG);
        q.im_w = __builtin_inff();
        CHECK(yp_detect_yolov3(hs, 3, &q, ff, 10, fi, nullptr, fake, big, nullptr) == YP_ERR_ARG);
        q.im_w = 1920.0f;                              // decoded rows carry no input size
        CHECK(yp_write_results(ff, 10, &q, ff, 10, fi, nullptr, fake, big, nullptr) == YP_ERR_ARG);
        q = p; q.mode = YP_MODE_REFERENCE;
        CHECK(yp_detect_yolov3(hs, 3, &q, ff, 10, fi, nullptr, fake, big, nullptr) == YP_ERR_ARG);
        q = p; q.batch = 0;
        CHECK(yp_detect_yolov3(hs, 3, &q, ff, 10, fi, nullptr, fake, big, nullptr) == YP_ERR_ARG);
        q = p; q.batch = 70000;
        CHECK(yp_detect_yolov3(hs, 3, &q, ff, 10, fi, nullptr, fake, big, nullptr) == YP_ERR_ARG);
        q = p; q.num_classes = 5000;
        CHECK(yp_detect_yolov3(hs, 3, &q, ff, 10, fi, nullptr, fake, big, nullptr) == YP_ERR_ARG);
        q = p; q.nms_thr = __builtin_nanf("");
        CHECK(yp_detect_yolov3(hs, 3, &q, ff, 10, fi, nullptr, fake, big, nullptr) == YP_ERR_ARG);
    }
    {
        YpHead bad[3] = {hs[0], hs[1], hs[2]};
        bad[1].num_anchors = YP_MAX_ANCHORS + 1;
        CHECK(yp_detect_yolov3(bad, 3, &p, ff, 10, fi, nullptr, fake, big, nullptr) == YP_ERR_UNSUPPORTED);
        bad[1] = hs[1]; bad[2].grid_w = 0;
        CHECK(yp_detect_yolov3(bad, 3, &p, ff, 10, fi, nullptr, fake, big, nullptr) == YP_ERR_ARG);
        bad[2] = hs[2]; bad[0].data = nullptr;
        CHECK(yp_detect_yolov3(bad, 3, &p, ff, 10, fi, nullptr, fake, big, nullptr) == YP_ERR_ARG);
        bad[0] = hs[0]; bad[0].grid_h = 5000; bad[0].grid_w = 5000;
        CHECK(yp_detect_yolov3(bad, 3, &p, ff, 10, fi, nullptr, fake, big, nullptr) == YP_ERR_UNSUPPORTED);
    }
    // several sets per launch
    CHECK(yp_detect_yolov3_multi(hs, 3, 0, &p, ff, 10, fi, nullptr, fake, big, nullptr) == YP_ERR_UNSUPPORTED);
    CHECK(yp_detect_yolov3_multi(hs, 3, YP_MAX_SETS + 1, &p, ff, 10, fi, nullptr, fake, big, nullptr) ==
          YP_ERR_UNSUPPORTED);
    // (any capacity keeps the sets' slabs 16-byte aligned: a row is 32 bytes; the base pointer must be)
    CHECK(yp_detect_yolov3_multi(hs, 3, 2, &p, ff + 1, 11, fi, nullptr, fake, big, nullptr) == YP_ERR_ARG);
    {
        YpHead two[6];
        std::memcpy(two, hs, sizeof two);
        two[4].grid_h = 27;
        CHECK(yp_detect_yolov3_multi(two, 3, 2, &p, ff, 10, fi, nullptr, fake, big, nullptr) == YP_ERR_ARG);
        two[4] = hs[1]; two[5].anchors[3] += 1.0f;
        CHECK(yp_detect_yolov3_multi(two, 3, 2, &p, ff, 10, fi, nullptr, fake, big, nullptr) == YP_ERR_ARG);
        const size_t one = yp_detect_yolov3_workspace_bytes(hs, 3, p.batch, 5);
        CHECK(yp_detect_yolov3_multi(hs, 3, 2, &p, ff, 10, fi, nullptr, fake, one, nullptr) == YP_ERR_WORKSPACE);
    }
    // the entry point that allocates nothing: too large a problem is refused before anything else
    {
        YpParams q = p;
        q.batch = 64;
        CHECK(yp_noalloc_scratch_bytes() == (size_t)YP_NOALLOC_SCRATCH_BYTES);
        CHECK(yp_detect_yolov3_noalloc(hs, 3, &q, ff, 10, fi, nullptr, nullptr) == YP_ERR_SCRATCH);
        CHECK(yp_detect_yolov3_noalloc(hs, 3, nullptr, ff, 10, fi, nullptr, nullptr) == YP_ERR_ARG);
        CHECK(yp_detect_yolov3_noalloc(nullptr, 3, &q, ff, 10, fi, nullptr, nullptr) == YP_ERR_ARG);
    }
    // decoded-rows and reference-mode entries
    CHECK(yp_write_results(nullptr, 10, &p, ff, 10, fi, nullptr, fake, big, nullptr) == YP_ERR_ARG);
    CHECK(yp_write_results(ff, 0, &p, ff, 10, fi, nullptr, fake, big, nullptr) == YP_ERR_ARG);
    CHECK(yp_write_results(ff, 10, &p, ff, 10, fi, nullptr, fake, 64, nullptr) == YP_ERR_WORKSPACE);
    {
        YpParams r;
        std::memset(&r, 0, sizeof r);
        r.mode = YP_MODE_REFERENCE; r.batch = 4; r.num_classes = 5; r.conf_thr = 0.6f; r.nms_thr = 0.5f;
        r.im_h = 1280.f; r.im_w = 1920.f;
        YpHead rh = head(13, 5, 0.f);
        const int bad_out[] = {0, -1, (1 << 20) + 1};
        for (int m : bad_out) {
            r.max_out = m;
            CHECK(yp_detect_reference(&rh, &r, ff, ff, fi, fi, fi, fi, fake, big, nullptr) == YP_ERR_ARG);
            CHECK(yp_filter_boxes_reference(ff, ff, ff, ff, 845, &r, ff, ff, fi, fi, fi, fi, fake, big, nullptr) ==
                  YP_ERR_ARG);
        }
        r.max_out = 20;
        CHECK(yp_detect_reference(nullptr, &r, ff, ff, fi, fi, fi, fi, fake, big, nullptr) == YP_ERR_ARG);
        CHECK(yp_detect_reference(&rh, &r, ff, ff, fi, fi, fi, fi, fake, 64, nullptr) == YP_ERR_WORKSPACE);
        rh.num_anchors = 9;
        CHECK(yp_detect_reference(&rh, &r, ff, ff, fi, fi, fi, fi, fake, big, nullptr) == YP_ERR_UNSUPPORTED);
        CHECK(yp_filter_boxes_reference(ff, ff, ff, ff, 0, &r, ff, ff, fi, fi, fi, fi, fake, big, nullptr) ==
              YP_ERR_ARG);
    }
    CHECK(yp_predict_transform_reference(nullptr, 1, 13, 13, 5, 5, ff, ff, ff, ff, ff, nullptr) == YP_ERR_ARG);
    CHECK(yp_predict_transform_yolov3(hs, 3, 0, 5, ff, nullptr) == YP_ERR_ARG);
    CHECK(yp_predict_transform_yolov3(hs, 9, 1, 5, ff, nullptr) == YP_ERR_ARG);
    CHECK(yp_bbox_iou(nullptr, ff, 3, 4, ff, nullptr) == YP_ERR_ARG);
    CHECK(yp_bbox_iou(ff, ff, 3, 2, ff, nullptr) == YP_ERR_ARG);
    CHECK(yp_bbox_iou(ff, ff, 0, 4, ff, nullptr) == YP_OK);
    CHECK(yp_profile_stop(nullptr, nullptr, nullptr, nullptr) == YP_ERR_ARG);
    CHECK(yp_profile_start(0) == YP_ERR_ARG);

    // validation statistics
    CHECK(yp_match_detections_workspace_bytes(1) == 256 && yp_match_detections_workspace_bytes(0) == 0);
    CHECK(yp_match_detections_workspace_bytes(((int64_t)1 << 30) + 1) == 0);
    CHECK(yp_match_detections(nullptr, fi, 10, ff, ff, fi, fi, 10, 5, 0.5f, fi, ff, fi, fake, big, nullptr) ==
          YP_ERR_ARG);
    CHECK(yp_match_detections(ff, fi, 0, ff, ff, fi, fi, 10, 5, 0.5f, fi, ff, fi, fake, big, nullptr) == YP_ERR_ARG);
    CHECK(yp_match_detections(ff, fi, 10, ff, ff, fi, fi, -3, 5, 0.5f, fi, ff, fi, fake, big, nullptr) == YP_ERR_ARG);
    CHECK(yp_match_detections(ff, fi, 10, ff, ff, fi, fi, 10, 4097, 0.5f, fi, ff, fi, fake, big, nullptr) ==
          YP_ERR_ARG);
    CHECK(yp_match_detections(ff, fi, 10, ff, ff, fi, fi, 10, 5, __builtin_nanf(""), fi, ff, fi, fake, big,
                              nullptr) == YP_ERR_ARG);
    CHECK(yp_match_detections(ff + 1, fi, 10, ff, ff, fi, fi, 10, 5, 0.5f, fi, ff, fi, fake, big, nullptr) ==
          YP_ERR_ARG);
    CHECK(yp_match_detections(ff, fi, 10, ff, ff, fi, fi, 10, 5, 0.5f, fi, nullptr, fi, fake, 8, nullptr) ==
          YP_ERR_WORKSPACE);

    if (g_fail) {
        std::printf("%d of %d checks failed\n", g_fail, g_checks);
        return 1;
    }
    std::printf("ok %d\n", g_checks);
    return 0;
}
