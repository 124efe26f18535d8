// yp_image.h -- greedy NMS of a WHOLE IMAGE by one workgroup (YOLOv3 mode: one segment per class).
// Included by yolopost.hip inside its anonymous namespace (it uses that file's wave helpers,
// SegArgs and yp_math.h).
//
// k_segment_nms gives every (image, class) segment a workgroup of its own, and every one of them
// scans all candidate tiles of the image to find its class.  For images with a few hundred to a
// couple of thousand candidates that scan, the fixed costs of a workgroup and the idle ends of its
// loops are most of the work.  Here one workgroup takes the image:
//
//   gather   the image's tiles are read ONCE; a candidate reserves a place in its class's region
//            of an LDS key pool with an LDS atomic (any order: the keys are sorted next);
//   sort     rank sort inside every class, one task per 64 elements, tasks dealt to the waves;
//   boxes    one trip to memory for the sorted boxes of all classes;
//   pairs    the triangular suppression bitmask of every class, in strips of 32 rows: half-word h
//            of column j holds the rows 32h..32h+31 that precede j and suppress it.  A strip's
//            columns are cut into units of eight; both halves of a wave hold the strip's 32 rows,
//            the lower half tests the first four columns of a unit and the upper half the other
//            four (a 64-row strip would idle half its lanes over the 64 columns of its diagonal
//            block, a 32-row strip over 32).  The units of all classes are shared out evenly
//            over the waves whatever the classes' sizes; two columns are tested per pass with
//            packed binary32 arithmetic (v_pk_add_f32 / v_pk_mul_f32 do two operations per lane
//            and instruction; every operation is the one yp_suppress_pretest performs, so the
//            verdicts are the same bits);
//   sweep    one wave per class walks its mask with the fixed-point iteration of k_segment_nms,
//            all classes at the same time, and writes the kept slots.
//
// Classes whose masks do not fit the pool together are handled in groups, one after the other.
// An image that does not fit (more candidates than the kernel's capacity, a class beyond its
// region, a box with x2 < x1 or a non-finite area) is flagged in img_flag and left to
// k_segment_nms, which runs after this kernel and skips the images this kernel finished.  Outputs are exactly the
// arrays k_segment_nms writes (kept_slot, seg_kept, seg_cand, seg_region, seg_pairs, seg_ties).
#pragma once

constexpr int kImgMaxClasses = 32;            // classes an image may have on this path
constexpr int kImgMaxTiles = 256;             // candidate tiles per image beyond which it is not tried
// Launches with fewer images stay with k_segment_nms: one workgroup per image lives about 33 us
// whatever the launch size, so it pays once the images fill the GPU (measured on an MI355X with
// 714 candidates per image: 256 images 42 us against 32 us for the segment kernel when the launch
// has the GPU to itself, 512 images 44 against 62; with two launches in flight the smaller
// instruction count of this kernel wins from 256 images on).
// The environment variable YP_IMG_MIN_IMAGES overrides it (the tests use 1 and a huge value to
// run every case on both paths).
constexpr int kImgMinImages = 256;
// Workgroups of kImgThreads threads, in two sizes (template arguments CAP = candidates of an image,
// POOL = 64-bit words for the gathered keys and later the mask words): the large one takes images
// of up to 1 728 candidates in 80 KB of LDS; the small one 1 024 candidates in 50 KB, so that two
// of its workgroups leave a third of a CU's LDS -- and, at 8 waves each, half its wave slots -- to
// the waves of another kernel: the front kernel of the next launch on a second stream, which is
// bound by HBM where this one is bound by instruction issue.  The host picks by the image size
// (more than kImgSmallTiles candidate tiles: the large one); an image beyond the chosen capacity
// goes to k_segment_nms as always.  Measured on an MI355X, 8 batches of 64 images at 416x416 per
// launch: with two launches in flight 1024-thread workgroups of either size 6.18 M images/s,
// 512-thread small ones 6.73 M; one launch at a time the NMS kernels take 40.7 us with 1024
// threads per image and 43.6 us with 512.  At 608x608 (large size): 2.19 M against 2.35 M.
constexpr int kImgThreads = 512;
constexpr int kImgCapLarge = 1728, kImgPoolLarge = 4096;
constexpr int kImgCapSmall = 1024, kImgPoolSmall = 2560;
constexpr int kImgSmallTiles = 40;            // (416x416: 30 tiles of 128 cells; 608x608: 61)
constexpr int kImgHalves = 2 * kSmallWords;   // 32-row strips of the largest class
constexpr int kImgStrips = kImgMaxClasses * kImgHalves;    // strip t = (class t / 16, row block t % 16)
constexpr int kImgUnit = 8;                   // columns per unit of the pair loop (four per half wave)
static_assert(kImgHalves == 16, "strip index arithmetic");
// 32-bit mask words of a class of `cap` candidates: sum over its strips of cap - 32 h
constexpr int img_mask_words(int cap) { return kImgHalves * cap - 16 * kImgHalves * (kImgHalves - 1); }

// dst (one value per lane) takes the wave-uniform value val in lane l (a literal) and keeps the
// rest.  (The CPU emulation the tests run these kernels under defines the macro itself.)
#ifndef YP_WRITELANE
#define YP_WRITELANE(dst, val, l) asm("v_writelane_b32 %0, %1, " #l : "+v"(dst) : "s"(val))
#endif

// Prefix record of a strip, four running sums packed into 64 bits: units of the pair loop (17
// bits), 32-bit mask words (19), candidates (16; counted at a class's first strip) and sort tasks
// (10; one per 64 candidates).  512 strips of at most 64 units and 512 words, 32 classes of at
// most 513 candidates fit.
__device__ __forceinline__ u64 img_pack(int units, int words, int cand, int tasks)
{
    return (u64)(unsigned)units | ((u64)(unsigned)words << 17) | ((u64)(unsigned)cand << 36) |
           ((u64)(unsigned)tasks << 52);
}
__device__ __forceinline__ int img_units(u64 p) { return (int)(p & 0x1ffffull); }
__device__ __forceinline__ int img_words(u64 p) { return (int)((p >> 17) & 0x7ffffull); }
__device__ __forceinline__ int img_cand(u64 p) { return (int)((p >> 36) & 0xffffull); }
__device__ __forceinline__ int img_tasks(u64 p) { return (int)(p >> 52); }

__device__ __forceinline__ u64 wave_inclusive_scan64(u64 v, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u64 t = __shfl_up(v, off, 64);
        if (lane >= off) v += t;
    }
    return v;
}

template <int MODE, int NT, int CAP, int POOL>
__global__ __launch_bounds__(NT)
void k_image_nms(SegArgs s)
{
    constexpr int kImgThreads = NT, kImgWaves = NT / 64, kImgCap = CAP, kImgPool = POOL;
    static_assert(kImgStrips <= kImgThreads, "one thread per strip");
    __shared__ __align__(16) u64 s_pool[kImgPool];
    __shared__ __align__(16) float4 s_box[kImgCap + 8];   // (+8: the last unit of a class may read on)
    __shared__ float s_area[kImgCap + 8];
    __shared__ int s_slot[kImgCap];
    __shared__ u64 s_pref[kImgStrips + 1];     // packed sums over the strips before strip t
    __shared__ u64 s_wsum[kImgWaves];
    __shared__ int s_tasks[kImgStrips / 2];    // the strips that start a sort task
    __shared__ int s_cnt[kImgMaxClasses];      // candidates per class
    __shared__ unsigned int s_tiesc[kImgMaxClasses];
    __shared__ int s_flag;                     // the image is left to k_segment_nms
    __shared__ int s_ties_in;                  // bit c: class c has two candidates of equal objectness

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int b = blockIdx.x;
    [[maybe_unused]] const int stamp_block = b;   // (diagnostic builds)
    const int C = s.C, tpi = s.tpi, ts = s.ts;
    const size_t ibase = (size_t)b * s.spi;
    const int *tcount = s.tile_count + (size_t)b * tpi;
    // The pool as 32-bit words: class c owns [2 c cs, 2 (c + 1) cs) -- cs sort keys (the order
    // bits of the objectness, inverted: ascending key = descending objectness), then the cs slot
    // ids that belong to them; at most cc candidates, the 8 keys behind them are padding.
    constexpr int kClassCap = img_mask_words(kSmallCap) <= 2 * kImgPool ? kSmallCap : 384;
    static_assert(img_mask_words(kClassCap) <= 2 * kImgPool, "the mask of the largest class fits the pool");
    const int cs = (kImgPool / C) & ~7;
    const int cc = cs - 8 < kClassCap ? cs - 8 : kClassCap;
    unsigned int *pool32 = reinterpret_cast<unsigned int *>(s_pool);

    STAMP(0);
    if (tid < kImgMaxClasses) {
        s_cnt[tid] = 0;
        s_tiesc[tid] = 0u;
    }
    if (tid == 0) {
        s_flag = 0;
        s_ties_in = 0;
    }
    // (a sorted position nobody writes keeps its -1: two candidates of a class had equal keys)
    for (int i = tid; i < kImgCap; i += kImgThreads) s_slot[i] = -1;
    __syncthreads();

    // ---- gather: wave w reads tiles w, w + 8, ... four at a time; lane r looks at candidate r
    // of each (count, key and class are requested together, so a round is one trip to memory)
    {
        constexpr int kGatherTiles = 4;
        auto place = [&](bool valid, float kf, int cl, int slot) {
            if (valid) {
                const int pos = atomicAdd(&s_cnt[cl], 1);
                if (pos < cc) {
                    pool32[2 * cl * cs + pos] = ~yp_float_order_bits(kf);
                    pool32[2 * cl * cs + cs + pos] = (unsigned int)slot;
                }
            }
        };
        for (int t0 = wv; t0 < tpi; t0 += kImgWaves * kGatherTiles) {
            int cnt[kGatherTiles], cl[kGatherTiles];
            float kf[kGatherTiles];
#pragma unroll
            for (int u = 0; u < kGatherTiles; ++u) {
                const int t = t0 + u * kImgWaves;
                const int tile = t < tpi ? t : tpi - 1;
                const int slot = tile * ts + lane;
                cnt[u] = t < tpi ? tcount[tile] : 0;
                kf[u] = s.cand_key[ibase + slot];
                cl[u] = s.cand_cls[ibase + slot];
            }
#pragma unroll
            for (int u = 0; u < kGatherTiles; ++u)
                place(lane < cnt[u], kf[u], cl[u], (t0 + u * kImgWaves) * ts + lane);
            // tiles with more than 64 candidates (rare): the rest, 64 at a time
#pragma unroll
            for (int u = 0; u < kGatherTiles; ++u) {
                for (int r0 = 64; r0 < cnt[u]; r0 += 64) {
                    const int r = r0 + lane;
                    const int slot = (t0 + u * kImgWaves) * ts + (r < cnt[u] ? r : 0);
                    place(r < cnt[u], s.cand_key[ibase + slot], s.cand_cls[ibase + slot], slot);
                }
            }
        }
    }
    __syncthreads();
    STAMP(1);

    // ---- strip table: thread t <-> strip (class t / 16, row block t % 16); a strip holds the
    // columns 32h .. K_c - 1 of its class against the rows 32h .. 32h + 31
    {
        u64 v = 0ull;
        int ncol = 0;
        bool task = false;
        if (tid < kImgStrips) {
            const int c = tid >> 4, h = tid & 15;
            int kc = c < C ? s_cnt[c] : 0;
            if (kc > cc) {
                atomicOr(&s_flag, 1);
                kc = cc;
            }
            ncol = kc - (h << 5);
            task = ncol > 0 && (h & 1) == 0;   // (a sort task: 64 candidates from 32h on)
            if (ncol > 0) v = img_pack((ncol + kImgUnit - 1) / kImgUnit, ncol, h == 0 ? kc : 0, task ? 1 : 0);
            if (h == 0)   // padding for the rank sort, which reads its keys eight at a time
                for (int i = kc; i < ((kc + 7) & ~7); ++i) pool32[2 * c * cs + i] = ~0u;
        }
        const u64 incl = wave_inclusive_scan64(v, lane);
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        u64 woff = 0ull;
        for (int w = 0; w < kImgWaves; ++w) woff += w < wave ? s_wsum[w] : 0ull;
        const u64 excl = woff + incl - v;
        if (tid < kImgStrips) {
            s_pref[tid] = excl;
            if (task) s_tasks[img_tasks(excl)] = tid;
            if (tid == kImgStrips - 1) s_pref[kImgStrips] = excl + v;
        }
    }
    __syncthreads();
    STAMP(2);
    const u64 p_all = s_pref[kImgStrips];
    const int k_img = img_cand(p_all);
    if (s_flag != 0 || k_img > kImgCap) {   // (uniform) does not fit: k_segment_nms takes the image
        if (tid == 0) s.img_flag[b] = 1;
        return;
    }
    const int max_out = s.max_out > 0 ? s.max_out : 0x7fffffff;
    const YpThr thr = yp_make_thr(s.thr);

    // ---- rank sort inside every class: a task is 64 elements of one class against all its keys.
    // Only the 32 key bits are compared at first.  Equal objectness values inside a class have to
    // be ordered by slot; compared by key alone they get the same rank, some position stays
    // unwritten (still -1), and the sort is repeated with the slot as the tie-break -- for about
    // one image in a hundred at 140 candidates per class (the keys are binary32 values in (0.5, 1)).
    auto rank_sort = [&](const bool exact, const unsigned int classes) {
        const int n_tasks = img_tasks(p_all);
        for (int k = wv; k < n_tasks; k += kImgWaves) {
            const int t = __builtin_amdgcn_readfirstlane(s_tasks[k]);
            const int c = t >> 4, w = (t & 15) >> 1;
            if (!((classes >> c) & 1u)) continue;
            const int kc = __builtin_amdgcn_readfirstlane(s_cnt[c]);
            const int base = img_cand(s_pref[c << 4]);
            const unsigned int *in = pool32 + 2 * c * cs;
            const int e = (w << 6) + lane;
            const unsigned int mine = e < kc ? in[e] : 0u;
            const unsigned int my_slot = e < kc ? in[cs + e] : 0u;
            const int4 *in4 = reinterpret_cast<const int4 *>(in);
            const int4 *sl4 = reinterpret_cast<const int4 *>(in + cs);
            const int kp = (kc + 7) & ~7;
            int rank = 0;
            if (!exact) {
                for (int j = 0; j < kp; j += 8) {
                    const int4 q0 = in4[j >> 2], q1 = in4[(j >> 2) + 1];
                    rank += ((unsigned int)q0.x < mine) + ((unsigned int)q0.y < mine) +
                            ((unsigned int)q0.z < mine) + ((unsigned int)q0.w < mine);
                    rank += ((unsigned int)q1.x < mine) + ((unsigned int)q1.y < mine) +
                            ((unsigned int)q1.z < mine) + ((unsigned int)q1.w < mine);
                }
            } else {
                // (key, slot) pairs in lexicographic order; the padding keys are larger than any key,
                // whatever the slot words behind the class's candidates hold
                auto before = [&](int kj, int sj) {
                    return ((unsigned int)kj < mine) | (((unsigned int)kj == mine) & ((unsigned int)sj < my_slot));
                };
                for (int j = 0; j < kp; j += 4) {
                    const int4 q = in4[j >> 2], r = sl4[j >> 2];
                    rank += before(q.x, r.x) + before(q.y, r.y) + before(q.z, r.z) + before(q.w, r.w);
                }
            }
            if (e < kc) s_slot[base + rank] = (int)my_slot;
        }
    };
    rank_sort(false, ~0u);
    __syncthreads();
    for (int i = tid; i < k_img; i += kImgThreads) {
        if (s_slot[i] < 0) {            // the class that owns position i had two equal keys
            int c = 0;
            while (c + 1 < C && img_cand(s_pref[(c + 1) << 4]) <= i) ++c;
            atomicOr(&s_ties_in, (int)(1u << c));
        }
    }
    __syncthreads();
    if (s_ties_in != 0) {               // (uniform) those classes once more, the slot as tie-break
        rank_sort(true, (unsigned int)s_ties_in);
        __syncthreads();
    }
    STAMP(3);

    // ---- the sorted boxes of all classes, prepared for the pair tests; the packed test needs
    // x2 >= x1, y2 >= y1 and a finite area (then every union is >= 1)
    {
        bool safe = true;
        for (int i = tid; i < k_img + 8; i += kImgThreads) {
            float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
            float area = 1.0f;
            if (i < k_img) {
                const int slot = s_slot[i];
                safe = safe && slot >= 0;
                const float4 raw = s.cand_box[ibase + (slot >= 0 ? slot : 0)];
                const YpBox q = {raw.x, raw.y, raw.z, raw.w};
                const YpPrep pp = yp_prep<MODE>(q);
                bx = make_float4(pp.lo0, pp.lo1, pp.hi0, pp.hi1);
                area = pp.area;
                // (comparisons that a NaN fails)
                safe = safe && (pp.hi0 >= pp.lo0) && (pp.hi1 >= pp.lo1) && (area < 1e30f);
            }
            s_box[i] = bx;
            s_area[i] = area;
        }
        if (!safe) atomicOr(&s_flag, 1);
    }
    __syncthreads();
    if (s_flag != 0) {
        if (tid == 0) s.img_flag[b] = 1;
        return;
    }
    if (tid == 0) s.img_flag[b] = 0;
    STAMP(4);

    // ---- groups of consecutive classes whose masks fit the pool together
    for (int g0 = 0; g0 < C;) {
        const u64 p0 = s_pref[g0 << 4];
        int g1 = g0 + 1;
        while (g1 < C && img_words(s_pref[(g1 + 1) << 4]) - img_words(p0) <= 2 * kImgPool) ++g1;
        const u64 p1 = s_pref[g1 << 4];
        const int m0 = img_words(p0);

        // -- pairs: the group's units in equal contiguous shares, one per wave
        {
            const int u_first = img_units(p0), n_units = img_units(p1) - u_first;
            int u = u_first + (int)((long long)n_units * wv / kImgWaves);
            const int u_hi = u_first + (int)((long long)n_units * (wv + 1) / kImgWaves);
            unsigned int ties = 0u;
            int ties_class = -1;
            const int half = lane >> 5, hl = lane & 31;   // lower / upper half of the wave, lane inside it
            if (u < u_hi) {
                // the last strip whose prefix does not exceed u (empty strips share a prefix with
                // their successor, so the search lands on the one that holds the unit)
                int lo = g0 << 4, hi = g1 << 4;
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (img_units(s_pref[mid]) <= u) lo = mid;
                    else hi = mid;
                }
                int t = lo;
                while (u < u_hi) {
                    while (img_units(s_pref[t + 1]) <= u) ++t;
                    t = __builtin_amdgcn_readfirstlane(t);
                    const int c = t >> 4, h = t & 15;
                    const u64 pt = s_pref[t];
                    const int kc = __builtin_amdgcn_readfirstlane(s_cnt[c]);
                    const int bs = __builtin_amdgcn_readfirstlane(img_cand(s_pref[c << 4]));
                    const int ub = __builtin_amdgcn_readfirstlane(img_units(pt));
                    int ue = __builtin_amdgcn_readfirstlane(img_units(s_pref[t + 1]));
                    ue = ue < u_hi ? ue : u_hi;
                    unsigned int *hw = pool32 + (img_words(pt) - m0);   // hw[j - 32h], j >= 32h
                    if (ties_class != c) {
                        if (ties_class >= 0 && ties != 0u && lane == 0) atomicAdd(&s_tiesc[ties_class], ties);
                        ties = 0u;
                        ties_class = c;
                    }
                    // both halves of the wave hold the strip's rows; the upper half works four
                    // columns ahead of the lower one
                    const int i = (h << 5) + hl;
                    const float4 ri = s_box[bs + (i < kc ? i : kc - 1)];
                    const float ai = s_area[bs + (i < kc ? i : kc - 1)];
                    const int n_rows = kc - (h << 5);           // rows, and columns, of the strip
                    const unsigned int rows_valid = n_rows >= 32 ? ~0u : (1u << n_rows) - 1u;
                    const float4 *cbase = s_box + bs + (h << 5) + 4 * half;
                    const float *abase = s_area + bs + (h << 5) + 4 * half;
                    // the rows of this strip that precede its column j (relative), none past the class
                    auto rows_before = [&](int j) {
                        return j >= n_rows ? 0u : (j >= 32 ? rows_valid : (1u << j) - 1u);
                    };
                    for (; u < ue; ++u) {
                        const int jr = kImgUnit * (u - ub);         // first column of the unit
                        float4 col[4];
                        float car[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            col[q] = cbase[jr + q];
                            car[q] = abase[jr + q];
                        }
                        u64 word[4];   // bits 0..31: column jr + q, bits 32..63: column jr + 4 + q
                        if (jr >= 32 && jr + kImgUnit <= n_rows) {
                            // past the diagonal block and inside the class: every row counts
                            const u64 rv = (u64)rows_valid | ((u64)rows_valid << 32);
#pragma unroll
                            for (int q = 0; q < 4; q += 2)
                                pair2_words<MODE>(ri, ai, col[q], col[q + 1], car[q], car[q + 1], thr, rv, rv,
                                                word[q], word[q + 1], ties);
                        } else {
#pragma unroll
                            for (int q = 0; q < 4; q += 2) {
                                const u64 ra = (u64)rows_before(jr + q) | ((u64)rows_before(jr + 4 + q) << 32);
                                const u64 rb = (u64)rows_before(jr + q + 1) | ((u64)rows_before(jr + 5 + q) << 32);
                                pair2_words<MODE>(ri, ai, col[q], col[q + 1], car[q], car[q + 1], thr, ra, rb,
                                                word[q], word[q + 1], ties);
                            }
                        }
                        // lanes 0..3 store the half-words of the columns jr .. jr + 3, lanes
                        // 32..35 those of jr + 4 .. jr + 7: one store instead of eight (the words
                        // are uniform over the wave; v_writelane puts one into one lane)
                        int mv = 0;
                        YP_WRITELANE(mv, (int)(unsigned int)word[0], 0);
                        YP_WRITELANE(mv, (int)(unsigned int)(word[0] >> 32), 32);
                        YP_WRITELANE(mv, (int)(unsigned int)word[1], 1);
                        YP_WRITELANE(mv, (int)(unsigned int)(word[1] >> 32), 33);
                        YP_WRITELANE(mv, (int)(unsigned int)word[2], 2);
                        YP_WRITELANE(mv, (int)(unsigned int)(word[2] >> 32), 34);
                        YP_WRITELANE(mv, (int)(unsigned int)word[3], 3);
                        YP_WRITELANE(mv, (int)(unsigned int)(word[3] >> 32), 35);
                        const int jc = jr + 4 * half + hl;
                        if (hl < 4 && jc < n_rows) hw[jc] = (unsigned int)mv;
                    }
                }
                if (ties_class >= 0 && ties != 0u && lane == 0) atomicAdd(&s_tiesc[ties_class], ties);
            }
        }
        __syncthreads();
        STAMP(5);

        // -- sweep: one wave per class of the group, pure bit operations on the mask
        for (int c = g0 + wv; c < g1; c += kImgWaves) {
            const int kc = s_cnt[c];
            const int bs = img_cand(s_pref[c << 4]);
            const int nwords = (kc + 63) >> 6;
            const size_t reg = ibase + bs;
            int nkept = 0;
            u64 keptw[kSmallWords];
#pragma unroll
            for (int w = 0; w < kSmallWords; ++w) keptw[w] = 0ull;
            // rows 64w .. 64w + 63 that suppress column j: the half-words of the strips 2w and 2w + 1
            // (the second one starts at column 64w + 32)
            auto mask_word = [&](int w, int j) {
                const int jrel = j - (w << 6);
                const unsigned int lo = pool32[img_words(s_pref[(c << 4) + 2 * w]) - m0 + jrel];
                const unsigned int hi =
                    jrel >= 32 ? pool32[img_words(s_pref[(c << 4) + 2 * w + 1]) - m0 + jrel - 32] : 0u;
                return ((u64)hi << 32) | (u64)lo;
            };
#pragma unroll
            for (int cwd = 0; cwd < kSmallWords; ++cwd) {
                if (cwd < nwords && nkept < max_out) {
                    const int j = (cwd << 6) + lane;
                    const int jj = j < kc ? j : (cwd << 6);
                    bool alive = j < kc;
#pragma unroll
                    for (int w = 0; w < cwd; ++w) alive = alive && (mask_word(w, jj) & keptw[w]) == 0ull;
                    const u64 mycol = (j < kc) ? mask_word(cwd, jj) : 0ull;
                    // greedy keep set of the chunk as the fixed point of
                    // keep[i] = alive[i] && no kept earlier row of the chunk suppresses i
                    u64 keep = __ballot(alive);
                    for (int it = 0; it < 66; ++it) {
                        const u64 nk = __ballot(alive && (mycol & keep) == 0ull);
                        if (nk == keep) break;
                        keep = nk;
                    }
                    const int room = max_out - nkept;
                    if (__popcll(keep) > room)
                        keep = __ballot(((keep >> lane) & 1ull) && __popcll(keep & lanes_below(lane)) < room);
                    keptw[cwd] = keep;
                    if ((keep >> lane) & 1ull)
                        s.kept_slot[reg + nkept + __popcll(keep & lanes_below(lane))] = s_slot[bs + j];
                    nkept += __popcll(keep);
                }
            }
            if (lane == 0) {
                const int seg = b * C + c;
                s.seg_cand[seg] = kc;
                s.seg_region[seg] = bs;
                s.seg_kept[seg] = nkept;
                s.seg_pairs[seg] = (unsigned int)((long long)kc * (kc - 1) / 2);
                s.seg_ties[seg] = s_tiesc[c];
            }
        }
        g0 = g1;
        if (g0 < C) __syncthreads();   // the next group's masks overwrite the pool
    }
    STAMP(6);
}
