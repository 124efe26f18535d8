// yp_math.h -- scalar binary32 arithmetic shared by every kernel of libyolopost.
//
// Everything here is a fixed sequence of IEEE-754 binary32 operations (floor, fma,
// multiply, add, subtract, correctly rounded divide, comparisons).  Compiled with
// -ffp-contract=off the GPU evaluates each function to the same bits as any conforming
// CPU, which is what makes kept-box index sets reproducible.  The functions are
// __host__ __device__ so a host-only build (tests/test_host_math.py compiles this header
// with g++) can be compared against the CPU oracle without a GPU.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define YP_HD __host__ __device__ __forceinline__
#else
#define YP_HD static inline
#endif

YP_HD float yp_bits2f(uint32_t u) { return __builtin_bit_cast(float, u); }
YP_HD uint32_t yp_f2bits(float f) { return __builtin_bit_cast(uint32_t, f); }

// exp(x): k = floor(x*log2(e) + 0.5), r = x - k*ln2 (two-term ln2), degree-5 minimax tail
// in Horner form with fma (Cephes expf coefficients), scaled by 2^k in two exact steps so
// that both factors are normal numbers.  Max error 0.98 ulp on [-104, 88.7].  Written
// without branches: the argument is clamped into the finite range for the arithmetic and
// the overflow / underflow / NaN results are selected at the end (in-range arguments go
// through exactly the operations listed above).
YP_HD float yp_exp(float x)
{
    const float hi = 88.72283935546875f, lo = -103.97207641601562f;
    float xc = x > hi ? hi : x;
    xc = xc < lo ? lo : xc;
    float kf = __builtin_floorf(__builtin_fmaf(xc, 1.44269504088896341f, 0.5f));
    float r = __builtin_fmaf(kf, -0.693359375f, xc);
    r = __builtin_fmaf(kf, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    p = __builtin_fmaf(p, r2, r);
    p = p + 1.0f;
    int k = (x == x) ? (int)kf : 0;
    int k1 = k >> 1;
    int k2 = k - k1;
    float s1 = yp_bits2f((uint32_t)(k1 + 127) << 23);
    float s2 = yp_bits2f((uint32_t)(k2 + 127) << 23);
    float y = (p * s1) * s2;
    y = x > hi ? __builtin_inff() : y;
    y = x < lo ? 0.0f : y;
    return (x == x) ? y : x;
}

YP_HD float yp_sigmoid(float x) { return 1.0f / (1.0f + yp_exp(-x)); }

// Two properties of yp_sigmoid that hold for EVERY binary32 argument and are checked
// exhaustively by tests/test_sigmoid_order.py (all 2.2e9 floats of [-104, 89]; outside that
// range the function is constant 0 or 1):
//   (1) it is monotone non-decreasing;
//   (2) for x in [kYpGapLo, kYpGapHi]:  yp_sigmoid(fl(x - kYpGap)) < yp_sigmoid(x).
// Together they let a kernel find "the first index of the largest sigmoid" from the raw values:
// with xm the largest argument (first index among equals), any argument <= fl(xm - kYpGap) has
// a strictly smaller sigmoid, so if no other argument lies in (fl(xm - kYpGap), xm) the answer
// is that index and yp_sigmoid(xm), computed once.
constexpr float kYpGap = 1e-3f, kYpGapLo = -80.0f, kYpGapHi = 8.0f;

// yp_exp of N independent arguments, written stage by stage so that neighbouring instructions
// belong to different arguments (a single wave then issues back to back instead of waiting
// for each dependent result).  Every element goes through exactly the operations of yp_exp,
// in the same order, so y[i] == yp_exp(x[i]) bit for bit.
template <int N>
YP_HD void yp_exp_staged(const float *x, float *y)
{
    const float hi = 88.72283935546875f, lo = -103.97207641601562f;
    float xc[N], kf[N], r[N], p[N], r2[N];
#define YP_STAGE for (int i = 0; i < N; ++i)
#pragma unroll
    YP_STAGE xc[i] = x[i] > hi ? hi : x[i];
#pragma unroll
    YP_STAGE xc[i] = xc[i] < lo ? lo : xc[i];
#pragma unroll
    YP_STAGE kf[i] = __builtin_floorf(__builtin_fmaf(xc[i], 1.44269504088896341f, 0.5f));
#pragma unroll
    YP_STAGE r[i] = __builtin_fmaf(kf[i], -0.693359375f, xc[i]);
#pragma unroll
    YP_STAGE r[i] = __builtin_fmaf(kf[i], 2.12194440e-4f, r[i]);
#pragma unroll
    YP_STAGE p[i] = __builtin_fmaf(1.9875691500e-4f, r[i], 1.3981999507e-3f);
#pragma unroll
    YP_STAGE p[i] = __builtin_fmaf(p[i], r[i], 8.3334519073e-3f);
#pragma unroll
    YP_STAGE p[i] = __builtin_fmaf(p[i], r[i], 4.1665795894e-2f);
#pragma unroll
    YP_STAGE p[i] = __builtin_fmaf(p[i], r[i], 1.6666665459e-1f);
#pragma unroll
    YP_STAGE p[i] = __builtin_fmaf(p[i], r[i], 5.0000001201e-1f);
#pragma unroll
    YP_STAGE r2[i] = r[i] * r[i];
#pragma unroll
    YP_STAGE p[i] = __builtin_fmaf(p[i], r2[i], r[i]);
#pragma unroll
    YP_STAGE p[i] = p[i] + 1.0f;
#pragma unroll
    YP_STAGE {
        const int k = (x[i] == x[i]) ? (int)kf[i] : 0;
        const int k1 = k >> 1;
        const int k2 = k - k1;
        const float s1 = yp_bits2f((uint32_t)(k1 + 127) << 23);
        const float s2 = yp_bits2f((uint32_t)(k2 + 127) << 23);
        float v = (p[i] * s1) * s2;
        v = x[i] > hi ? __builtin_inff() : v;
        v = x[i] < lo ? 0.0f : v;
        y[i] = (x[i] == x[i]) ? v : x[i];
    }
#undef YP_STAGE
}

// A box as four floats.  YOLOv3 mode: (x1,y1,x2,y2).  Reference mode: (y1,x1,y2,x2).
struct YpBox {
    float a, b, c, d;
};

// A box prepared for repeated IoU tests: corners (normalised by min/max in reference mode)
// and its area term.
struct YpPrep {
    float lo0, lo1, hi0, hi1, area;
};

// YOLOv3 tutorial bbox_iou, pixel-inclusive: area = (x2-x1+1)*(y2-y1+1).
YP_HD YpPrep yp_prep_v3(const YpBox &q)
{
    YpPrep p;
    p.lo0 = q.a; p.lo1 = q.b; p.hi0 = q.c; p.hi1 = q.d;
    p.area = (q.c - q.a + 1.0f) * (q.d - q.b + 1.0f);
    return p;
}

YP_HD float yp_iou_v3(const YpPrep &p, const YpPrep &q)
{
    const float i0 = p.lo0 > q.lo0 ? p.lo0 : q.lo0;
    const float i1 = p.lo1 > q.lo1 ? p.lo1 : q.lo1;
    const float j0 = p.hi0 < q.hi0 ? p.hi0 : q.hi0;
    const float j1 = p.hi1 < q.hi1 ? p.hi1 : q.hi1;
    float d0 = j0 - i0 + 1.0f;
    float d1 = j1 - i1 + 1.0f;
    if (!(d0 > 0.0f)) d0 = 0.0f;
    if (!(d1 > 0.0f)) d1 = 0.0f;
    const float inter = d0 * d1;
    return inter / (p.area + q.area - inter);
}

// tf.image.non_max_suppression IoU (the call at detect_net.py:136): corners normalised by
// min/max, no +1, a box of non-positive area has IoU 0 with everything.
YP_HD YpPrep yp_prep_tf(const YpBox &q)
{
    YpPrep p;
    p.lo0 = q.a < q.c ? q.a : q.c;
    p.lo1 = q.b < q.d ? q.b : q.d;
    p.hi0 = q.a > q.c ? q.a : q.c;
    p.hi1 = q.b > q.d ? q.b : q.d;
    p.area = (p.hi0 - p.lo0) * (p.hi1 - p.lo1);
    return p;
}

YP_HD float yp_iou_tf(const YpPrep &p, const YpPrep &q)
{
    if (p.area <= 0.0f || q.area <= 0.0f) return 0.0f;
    const float i0 = p.lo0 > q.lo0 ? p.lo0 : q.lo0;
    const float i1 = p.lo1 > q.lo1 ? p.lo1 : q.lo1;
    const float j0 = p.hi0 < q.hi0 ? p.hi0 : q.hi0;
    const float j1 = p.hi1 < q.hi1 ? p.hi1 : q.hi1;
    float d0 = j0 - i0;
    float d1 = j1 - i1;
    if (!(d0 > 0.0f)) d0 = 0.0f;
    if (!(d1 > 0.0f)) d1 = 0.0f;
    const float inter = d0 * d1;
    return inter / (p.area + q.area - inter);
}

// Does the earlier (higher-score) box p suppress the later box q?
//   YOLOv3 mode:    the later box survives only if IoU < thr   -> suppress = !(iou < thr)
//   reference mode: the later box is dropped when IoU > thr    -> suppress = iou > thr
template <int MODE>
YP_HD YpPrep yp_prep(const YpBox &q)
{
    return MODE == 0 ? yp_prep_v3(q) : yp_prep_tf(q);
}

template <int MODE>
YP_HD bool yp_suppresses_exact(const YpPrep &p, const YpPrep &q, float thr)
{
    if (MODE == 0) return !(yp_iou_v3(p, q) < thr);
    return yp_iou_tf(p, q) > thr;
}

// Threshold with guard bands for the division-free test below.
struct YpThr {
    float thr, lo, hi;
    bool fast;
};

YP_HD YpThr yp_make_thr(float thr)
{
    YpThr t;
    t.thr = thr;
    t.lo = thr * 0.99999904632568359375f;   // thr * (1 - 2^-20)
    t.hi = thr * 1.00000095367431640625f;   // thr * (1 + 2^-20)
    t.fast = thr > 1e-30f && thr < 1e30f;
    return t;
}

// Division-free pre-test, branch-free so that several independent tests overlap in the
// pipeline.  Returns 0 (does not suppress), 1 (suppresses) or 2 (undecided: the caller
// must run yp_suppresses_exact).  With r = inter/union (real), inter < fl(lo*union) implies
// r < thr*(1 - 2^-21), hence fl(r) < thr by monotonicity of rounding; inter > fl(hi*union)
// implies fl(r) > thr.  Undecided are only quotients within ~2^-20 of the threshold,
// non-finite or non-positive unions, and thresholds outside (1e-30, 1e30); so the final
// verdict equals the divide-and-compare one bit for bit.
template <int MODE>
YP_HD int yp_suppress_pretest(const YpPrep &p, const YpPrep &q, const YpThr &t)
{
    const float i0 = p.lo0 > q.lo0 ? p.lo0 : q.lo0;
    const float i1 = p.lo1 > q.lo1 ? p.lo1 : q.lo1;
    const float j0 = p.hi0 < q.hi0 ? p.hi0 : q.hi0;
    const float j1 = p.hi1 < q.hi1 ? p.hi1 : q.hi1;
    float d0 = MODE == 0 ? (j0 - i0 + 1.0f) : (j0 - i0);
    float d1 = MODE == 0 ? (j1 - i1 + 1.0f) : (j1 - i1);
    d0 = (d0 > 0.0f) ? d0 : 0.0f;
    d1 = (d1 > 0.0f) ? d1 : 0.0f;
    const float inter = d0 * d1;
    const float uni = p.area + q.area - inter;
    const bool yes = inter > t.hi * uni;
    const bool no = inter < t.lo * uni;
    // (no upper guard on the union: above 1e30 the bands decide "no", as the quotient does)
    const bool decided = t.fast & (uni > 1e-30f) & (yes | no);
    int v = decided ? (yes ? 1 : 0) : 2;
    if (MODE != 0) {
        const bool degenerate = (p.area <= 0.0f) | (q.area <= 0.0f);
        v = degenerate ? ((0.0f > t.thr) ? 1 : 0) : v;
    }
    return v;
}

template <int MODE>
YP_HD bool yp_suppresses(const YpPrep &p, const YpPrep &q, const YpThr &t)
{
    const int v = yp_suppress_pretest<MODE>(p, q, t);
    if (v == 2) return yp_suppresses_exact<MODE>(p, q, t.thr);
    return v == 1;
}

// Order-preserving map float -> uint32 (unsigned compare == float compare, NaN excluded).
YP_HD uint32_t yp_float_order_bits(float f)
{
    uint32_t u = yp_f2bits(f + 0.0f);  // +0.0f turns -0 into +0
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
