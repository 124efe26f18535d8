// yolopost.hip -- HIP kernels (gfx950, wave64) and the C ABI of libyolopost.so.
//
// Pipeline (three launches, no host synchronisation, no atomics on global memory):
//
//   front kernel   one of four loaders (raw YOLOv3 NCHW heads, raw reference head, decoded
//                  YOLOv3 rows, decoded reference 4-tuple) decodes each anchor, applies the
//                  confidence filter and compacts the survivors of its tile IN ANCHOR ORDER
//                  with wavefront ballots + popcount prefix; it writes the candidates
//                  (box, sort key, aux score, class) into the tile's slot range and the
//                  tile's survivor count.
//   k_segment_nms  one workgroup per NMS segment (image x class in YOLOv3 mode, image in
//                  reference mode): counts/gathers the segment's candidates, sorts them
//                  by (key descending, anchor order ascending) with a bitonic network in
//                  LDS, then runs greedy NMS in chunks of 64 sorted rows: a 64x64 IoU
//                  bitmask tile built by all waves with ballots, a wavefront sweep of
//                  that tile, and a column pass in which every thread tests the columns
//                  it owns against the chunk's kept rows staged in LDS.
//   k_gather_rows  (YOLOv3 mode) prefix over the segments' kept counts and the final
//                  (D, 8) row write.
//
// Semantics follow detect_net.py:90-183 of the reference in YP_MODE_REFERENCE and the
// YOLOv3 tutorial's predict_transform / write_results / bbox_iou in YP_MODE_YOLOV3; the
// per-function citations are in include/yolopost.h.
#include <hip/hip_runtime.h>

#include "yolopost.h"
#include "yp_math.h"

namespace {

typedef unsigned long long u64;

constexpr int kRowTile = 256;     // rows per tile in the row-mapped front kernels
constexpr int kSegThreads = 512;  // threads per NMS workgroup
constexpr int kSegWaves = kSegThreads / 64;
constexpr int kSortCap = 4096;    // largest segment sorted in LDS

// ------------------------------------------------------------------------ device structs

struct HeadGeom {
    const float *data;
    int gh, gw, A;
    int tile_begin;  // first tile of this head inside an image
    int row_begin;   // first output row (anchor index) of this head inside an image
    float sx, sy;
    float anc[2 * YP_MAX_ANCHORS];
};

struct HeadsArgs {
    HeadGeom h[YP_MAX_HEADS];
    int n_heads;
    int tpi;   // tiles per image
    int C;     // classes
    int N;     // rows per image over all heads
};

struct AnchorTable {
    float anc[2 * YP_MAX_ANCHORS];
};

// Candidate storage: slot (b, tile, r) lives at (b*tpi + tile)*ts + r; only r < count valid.
struct CandBuf {
    float4 *box;
    float *key;
    float *aux;
    int *cls;
    int *tile_count;
    int tpi;  // tiles per image
    int ts;   // slots per tile
};

struct SegArgs {
    int B, C, spi, tpi, ts, per_class, max_out, batch_offset;
    float thr;
    const float4 *cand_box;
    const float *cand_key;
    const float *cand_aux;
    const int *cand_cls;
    const int *tile_count;
    int *tile_prefix;        // [B*tpi] exclusive prefix of tile_count inside an image
    u64 *gkeys;              // [2*B*spi] sort area for segments larger than kSortCap
    float4 *sbox;            // [B*spi] boxes in sorted order, per segment region
    int *sslot;              // [B*spi] slot ids in sorted order
    unsigned char *removed;  // [B*spi] suppression flags in sorted order
    int *kept_slot;          // [B*spi] slot ids of kept rows, per segment region
    int *seg_kept;           // [n_seg]
    int *seg_cand;           // [n_seg]
    int *seg_region;         // [n_seg] region start inside the image
    unsigned int *seg_pairs; // [n_seg] IoU evaluations (low 32 bits)
    // reference-mode outputs, padded to max_out rows per image
    float *out_boxes;
    float *out_scores;
    int *out_classes;
    int *out_index;
    int *out_counts;
    int *surv_counts;
};

struct GatherArgs {
    int n_seg, C, spi, batch_offset;
    const float4 *cand_box;
    const float *cand_key;
    const float *cand_aux;
    const int *kept_slot;
    const int *seg_kept;
    const int *seg_cand;
    const int *seg_region;
    const unsigned int *seg_pairs;
    float *out_rows;
    long long cap;
    int *out_count;
    int *stats;
};

// --------------------------------------------------------------------------- wave helpers

__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;  // valid in lane 0
}

__device__ __forceinline__ int wave_inclusive_scan(int v, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(v, off, 64);
        if (lane >= off) v += t;
    }
    return v;
}

__device__ __forceinline__ u64 lanes_below(int lane) { return (1ull << lane) - 1ull; }

// Ordered rank of a surviving thread inside a tile whose threads are in row order
// (thread t <-> row t of the tile): ballot per wave, popcount prefix over waves.
__device__ __forceinline__ int tile_rank_rows(bool pass, u64 *s_bal, int &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = blockDim.x >> 6;
    const u64 bal = __ballot(pass);
    if (lane == 0) s_bal[wave] = bal;
    __syncthreads();
    int rank = __popcll(bal & lanes_below(lane));
    total = 0;
    for (int w = 0; w < nw; ++w) {
        const int c = __popcll(s_bal[w]);
        if (w < wave) rank += c;
        total += c;
    }
    return rank;
}

// Ordered rank when wave w holds anchor w of 64 consecutive cells (row = cell*A + anchor):
// rows of earlier cells first (any anchor), then earlier anchors of the same cell.
__device__ __forceinline__ int tile_rank_cells(bool pass, u64 *s_bal, int &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = blockDim.x >> 6;
    const u64 bal = __ballot(pass);
    if (lane == 0) s_bal[wave] = bal;
    __syncthreads();
    const u64 below = lanes_below(lane);
    int rank = 0;
    total = 0;
    for (int w = 0; w < nw; ++w) {
        const u64 m = s_bal[w];
        rank += __popcll(m & below);
        if (w < wave) rank += (int)((m >> lane) & 1ull);
        total += __popcll(m);
    }
    return rank;
}

__device__ __forceinline__ void store_cand(const CandBuf &cb, size_t slot, const YpBox &bx,
                                           float key, float aux, int cls)
{
    cb.box[slot] = make_float4(bx.a, bx.b, bx.c, bx.d);
    cb.key[slot] = key;
    cb.aux[slot] = aux;
    cb.cls[slot] = cls;
}

// ------------------------------------------------------------------ front: YOLOv3 heads
// Block = 64 cells x A anchors of one head of one image; wave w handles anchor w, so every
// channel load is 64 consecutive floats of one NCHW plane (256 B per wave instruction).
template <int CS>
__global__ void k_front_heads_v3(HeadsArgs ha, CandBuf cb, float conf)
{
    __shared__ u64 s_bal[YP_MAX_ANCHORS];
    const int tile = blockIdx.x, b = blockIdx.y;
    int h = 0;
#pragma unroll
    for (int i = 1; i < YP_MAX_HEADS; ++i)
        if (i < ha.n_heads && tile >= ha.h[i].tile_begin) h = i;
    const HeadGeom &g = ha.h[h];
    const int lane = threadIdx.x & 63, a = threadIdx.x >> 6;
    const int C = CS ? CS : ha.C;
    const int V = 5 + C;
    const int cells = g.gh * g.gw;
    const int cell = (tile - g.tile_begin) * 64 + lane;
    const bool active = (a < g.A) && (cell < cells);

    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f, t4 = 0.f;
    float tc[CS ? CS : 1];
    const float *p = g.data;
    if (active) {
        p = g.data + ((size_t)(b * g.A + a) * V) * cells + cell;
        t4 = p[(size_t)4 * cells];
        t0 = p[0];
        t1 = p[(size_t)1 * cells];
        t2 = p[(size_t)2 * cells];
        t3 = p[(size_t)3 * cells];
        if (CS) {
#pragma unroll
            for (int j = 0; j < (CS ? CS : 1); ++j) tc[j] = p[(size_t)(5 + j) * cells];
        }
    }
    float obj = 0.f;
    bool pass = false;
    if (active) {
        obj = yp_sigmoid(t4);
        pass = (obj > conf) && (obj != 0.0f);
    }
    int total;
    const int rank = tile_rank_cells(pass, s_bal, total);
    if (pass) {
        const int row = cell / g.gw, col = cell - row * g.gw;
        const float cx = (yp_sigmoid(t0) + (float)col) * g.sx;
        const float cy = (yp_sigmoid(t1) + (float)row) * g.sy;
        const float w = (yp_exp(t2) * g.anc[2 * a + 0]) * g.sx;
        const float hgt = (yp_exp(t3) * g.anc[2 * a + 1]) * g.sy;
        float best;
        int bi = 0;
        if (CS) {
            best = yp_sigmoid(tc[0]);
#pragma unroll
            for (int j = 1; j < (CS ? CS : 1); ++j) {
                const float s = yp_sigmoid(tc[j]);
                if (s > best) { best = s; bi = j; }
            }
        } else {
            best = yp_sigmoid(p[(size_t)5 * cells]);
            for (int j = 1; j < C; ++j) {
                const float s = yp_sigmoid(p[(size_t)(5 + j) * cells]);
                if (s > best) { best = s; bi = j; }
            }
        }
        const float hw = w * 0.5f, hh = hgt * 0.5f;
        YpBox bx = {cx - hw, cy - hh, cx + hw, cy + hh};
        const size_t slot = ((size_t)b * cb.tpi + tile) * cb.ts + rank;
        store_cand(cb, slot, bx, obj + 0.0f, best, bi);
    }
    if (threadIdx.x == 0) cb.tile_count[(size_t)b * cb.tpi + tile] = total;
}

// ---------------------------------------------------------------- front: reference head
// Thread t <-> row n = tile*256 + t of image b; a row is the V contiguous floats of one
// (row, col, anchor) (detect_net.py:153).
__global__ void k_front_grid_ref(const float *pred, int H, int W, int A, int C,
                                 AnchorTable at, CandBuf cb, float conf, float im_h,
                                 float im_w)
{
    __shared__ u64 s_bal[kRowTile / 64];
    const int tile = blockIdx.x, b = blockIdx.y;
    const int N = H * W * A, V = 5 + C;
    const int n = tile * kRowTile + threadIdx.x;
    bool pass = false;
    YpBox bx = {0.f, 0.f, 0.f, 0.f};
    float best = 0.f;
    int bi = 0;
    if (n < N) {
        const int a = n % A, k = n / A;
        const float gx = (float)(k % H), gy = (float)(k / H);   // detect_net.py:155-162
        const float *t = pred + ((size_t)b * N + n) * V;
        const float cx = (yp_sigmoid(t[0]) + gx) / (float)H;    // :166-168
        const float cy = (yp_sigmoid(t[1]) + gy) / (float)W;
        const float obj = yp_sigmoid(t[4]);                     // :171
        const float w = (yp_exp(t[2]) * at.anc[2 * a + 0]) / (float)H;  // :177
        const float hgt = (yp_exp(t[3]) * at.anc[2 * a + 1]) / (float)W;
        float m = t[5];                                         // softmax, :180
        for (int j = 1; j < C; ++j) m = t[5 + j] > m ? t[5 + j] : m;
        float s = 0.f;
        for (int j = 0; j < C; ++j) s = s + yp_exp(t[5 + j] - m);
        best = obj * (yp_exp(t[5] - m) / s);                    // :107-109
        for (int j = 1; j < C; ++j) {
            const float sc = obj * (yp_exp(t[5 + j] - m) / s);
            if (sc > best) { best = sc; bi = j; }
        }
        pass = best >= conf;                                    // :111
        const float hw = w * 0.5f, hh = hgt * 0.5f;             // :95-102, :133
        bx.a = (cy - hh) * im_h;
        bx.b = (cx - hw) * im_w;
        bx.c = (cy + hh) * im_h;
        bx.d = (cx + hw) * im_w;
    }
    int total;
    const int rank = tile_rank_rows(pass, s_bal, total);
    if (pass) store_cand(cb, ((size_t)b * cb.tpi + tile) * cb.ts + rank, bx, best, 0.f, bi);
    if (threadIdx.x == 0) cb.tile_count[(size_t)b * cb.tpi + tile] = total;
}

// ------------------------------------------------------- front: decoded reference 4-tuple
__global__ void k_front_split_ref(const float *center, const float *wh, const float *obj,
                                  const float *cls, int N, int C, CandBuf cb, float conf,
                                  float im_h, float im_w)
{
    __shared__ u64 s_bal[kRowTile / 64];
    const int tile = blockIdx.x, b = blockIdx.y;
    const int n = tile * kRowTile + threadIdx.x;
    bool pass = false;
    YpBox bx = {0.f, 0.f, 0.f, 0.f};
    float best = 0.f;
    int bi = 0;
    if (n < N) {
        const size_t o = (size_t)b * N + n;
        const float2 ce = reinterpret_cast<const float2 *>(center)[o];
        const float2 sz = reinterpret_cast<const float2 *>(wh)[o];
        const float ob = obj[o];
        const float *pc = cls + o * C;
        best = ob * pc[0];
        for (int j = 1; j < C; ++j) {
            const float sc = ob * pc[j];
            if (sc > best) { best = sc; bi = j; }
        }
        pass = best >= conf;
        const float hw = sz.x * 0.5f, hh = sz.y * 0.5f;
        bx.a = (ce.y - hh) * im_h;
        bx.b = (ce.x - hw) * im_w;
        bx.c = (ce.y + hh) * im_h;
        bx.d = (ce.x + hw) * im_w;
    }
    int total;
    const int rank = tile_rank_rows(pass, s_bal, total);
    if (pass) store_cand(cb, ((size_t)b * cb.tpi + tile) * cb.ts + rank, bx, best, 0.f, bi);
    if (threadIdx.x == 0) cb.tile_count[(size_t)b * cb.tpi + tile] = total;
}

// ------------------------------------------------------------ front: decoded YOLOv3 rows
__global__ void k_front_rows_v3(const float *rows, int N, int C, CandBuf cb, float conf)
{
    __shared__ u64 s_bal[kRowTile / 64];
    const int tile = blockIdx.x, b = blockIdx.y;
    const int V = 5 + C;
    const int n = tile * kRowTile + threadIdx.x;
    bool pass = false;
    YpBox bx = {0.f, 0.f, 0.f, 0.f};
    float best = 0.f, ob = 0.f;
    int bi = 0;
    if (n < N) {
        const float *p = rows + ((size_t)b * N + n) * V;
        ob = p[4];
        pass = (ob > conf) && (ob != 0.0f);
        if (pass) {
            const float hw = p[2] * 0.5f, hh = p[3] * 0.5f;
            bx.a = p[0] - hw;
            bx.b = p[1] - hh;
            bx.c = p[0] + hw;
            bx.d = p[1] + hh;
            best = p[5];
            for (int j = 1; j < C; ++j)
                if (p[5 + j] > best) { best = p[5 + j]; bi = j; }
        }
    }
    int total;
    const int rank = tile_rank_rows(pass, s_bal, total);
    if (pass) store_cand(cb, ((size_t)b * cb.tpi + tile) * cb.ts + rank, bx, ob + 0.0f, best, bi);
    if (threadIdx.x == 0) cb.tile_count[(size_t)b * cb.tpi + tile] = total;
}

// --------------------------------------------------------------------------- bitonic sort
// Ascending sort of n_pad (power of two) 64-bit keys by the whole workgroup.
template <class KeyPtr>
__device__ __forceinline__ void bitonic_sort(KeyPtr keys, int n_pad)
{
    const int half = n_pad >> 1;
    for (int k = 2; k <= n_pad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < half; t += kSegThreads) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const u64 x = keys[i], y = keys[l];
                const bool up = (i & k) == 0;
                if ((x > y) == up) {
                    keys[i] = y;
                    keys[l] = x;
                }
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------ segment NMS
template <int MODE>
__global__ __launch_bounds__(kSegThreads) void k_segment_nms(SegArgs s)
{
    __shared__ u64 s_keys[kSortCap];
    __shared__ float4 s_cbox[64];
    __shared__ YpPrep s_kprep[64];
    __shared__ u64 s_sup[64];
    __shared__ u64 s_mask[2];  // [0] alive rows of the chunk, [1] kept rows of the chunk
    __shared__ int s_cnt[4];   // [0] classes below, [1] this class, [2] gather cursor
    __shared__ int s_scan[kSegWaves];
    __shared__ unsigned int s_pairs;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int seg = blockIdx.x;
    const int b = s.per_class ? seg / s.C : seg;
    const int c = s.per_class ? seg - b * s.C : 0;
    const size_t ibase = (size_t)b * s.spi;
    const int *tcount = s.tile_count + (size_t)b * s.tpi;
    const int tpi = s.tpi, ts = s.ts;

    if (tid < 4) s_cnt[tid] = 0;
    if (tid == 0) s_pairs = 0u;
    __syncthreads();

    // ---- pass 1: how many candidates of this image have a lower class / this class
    {
        int lt = 0, eq = 0;
        for (int tb = wave * 64; tb < tpi; tb += kSegWaves * 64) {
            const int mycnt = (tb + lane < tpi) ? tcount[tb + lane] : 0;
            if (!s.per_class) {
                eq += mycnt;
                continue;
            }
            if (__ballot(mycnt > 0) == 0ull) continue;
            const int lim = (tpi - tb) < 64 ? (tpi - tb) : 64;
            for (int i = 0; i < lim; ++i) {
                const int cnt = __shfl(mycnt, i, 64);
                for (int r = lane; r < cnt; r += 64) {
                    const int cl = s.cand_cls[ibase + (size_t)(tb + i) * ts + r];
                    lt += (cl < c) ? 1 : 0;
                    eq += (cl == c) ? 1 : 0;
                }
            }
        }
        lt = wave_sum(lt);
        eq = wave_sum(eq);
        if (lane == 0) {
            if (lt) atomicAdd(&s_cnt[0], lt);
            if (eq) atomicAdd(&s_cnt[1], eq);
        }
    }
    __syncthreads();
    const int K = s_cnt[1];
    const int region = s_cnt[0];
    const size_t reg = ibase + region;
    if (tid == 0) {
        s.seg_cand[seg] = K;
        s.seg_region[seg] = region;
    }

    // reference mode reports positions in the image's compacted survivor list, which needs
    // the exclusive prefix of the tile counts
    if (MODE == YP_MODE_REFERENCE) {
        int carry = 0;
        for (int base = 0; base < tpi; base += kSegThreads) {
            const int v = (base + tid < tpi) ? tcount[base + tid] : 0;
            const int incl = wave_inclusive_scan(v, lane);
            if (lane == 63) s_scan[wave] = incl;
            __syncthreads();
            int woff = 0, total = 0;
            for (int w = 0; w < kSegWaves; ++w) {
                if (w < wave) woff += s_scan[w];
                total += s_scan[w];
            }
            if (base + tid < tpi) s.tile_prefix[(size_t)b * tpi + base + tid] = carry + woff + incl - v;
            carry += total;
            __syncthreads();
        }
    }

    if (K == 0) {
        if (tid == 0) {
            s.seg_kept[seg] = 0;
            s.seg_pairs[seg] = 0u;
            if (MODE == YP_MODE_REFERENCE) {
                s.out_counts[b] = 0;
                s.surv_counts[b] = 0;
            }
        }
        return;
    }

    // ---- pass 2: gather sort keys (key descending, slot ascending)
    const bool in_lds = K <= kSortCap;
    u64 *gk = s.gkeys + 2 * reg;
    int n_pad = 2;
    while (n_pad < K) n_pad <<= 1;
    for (int tb = wave * 64; tb < tpi; tb += kSegWaves * 64) {
        const int mycnt = (tb + lane < tpi) ? tcount[tb + lane] : 0;
        if (__ballot(mycnt > 0) == 0ull) continue;
        const int lim = (tpi - tb) < 64 ? (tpi - tb) : 64;
        for (int i = 0; i < lim; ++i) {
            const int cnt = __shfl(mycnt, i, 64);
            for (int r = lane; r < cnt; r += 64) {
                const int slot = (tb + i) * ts + r;
                if (s.per_class && s.cand_cls[ibase + slot] != c) continue;
                const u64 key = ((u64)(~yp_float_order_bits(s.cand_key[ibase + slot])) << 32) |
                                (u64)(unsigned int)slot;
                const int pos = atomicAdd(&s_cnt[2], 1);
                if (in_lds) s_keys[pos] = key;
                else gk[pos] = key;
            }
        }
    }
    for (int i = K + tid; i < n_pad; i += kSegThreads) {
        if (in_lds) s_keys[i] = ~0ull;
        else gk[i] = ~0ull;
    }
    __syncthreads();

    // ---- sort, then lay the segment out in sorted order
    if (in_lds) bitonic_sort(s_keys, n_pad);
    else bitonic_sort(gk, n_pad);
    for (int i = tid; i < K; i += kSegThreads) {
        const int slot = (int)(unsigned int)((in_lds ? s_keys[i] : gk[i]) & 0xffffffffull);
        s.sslot[reg + i] = slot;
        s.sbox[reg + i] = s.cand_box[ibase + slot];
        s.removed[reg + i] = 0;
    }
    __syncthreads();

    // ---- greedy NMS over chunks of 64 sorted rows
    const int max_out = s.max_out > 0 ? s.max_out : 0x7fffffff;
    const float thr = s.thr;
    int nkept = 0;
    unsigned int pairs = 0;
    for (int r0 = 0; r0 < K && nkept < max_out; r0 += 64) {
        if (wave == 0) {
            const int i = r0 + lane;
            const bool valid = i < K;
            float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
            bool alive = false;
            if (valid) {
                bx = s.sbox[reg + i];
                alive = s.removed[reg + i] == 0;
            }
            s_cbox[lane] = bx;
            const u64 am = __ballot(alive);
            if (lane == 0) s_mask[0] = am;
        }
        __syncthreads();
        const u64 alive_mask = s_mask[0];
        if (alive_mask == 0ull) {
            __syncthreads();
            continue;
        }
        // 64x64 suppression bitmask tile: wave w builds rows w, w+8, ...; lane = column
        {
            const float4 cj = s_cbox[lane];
            const YpBox bj = {cj.x, cj.y, cj.z, cj.w};
            const YpPrep pj = yp_prep<MODE>(bj);
            for (int l = wave; l < 64; l += kSegWaves) {
                u64 m = 0ull;
                if ((alive_mask >> l) & 1ull) {
                    const float4 cl = s_cbox[l];
                    const YpBox bl = {cl.x, cl.y, cl.z, cl.w};
                    const YpPrep pl = yp_prep<MODE>(bl);
                    const bool sup = (lane > l) && ((alive_mask >> lane) & 1ull) &&
                                     yp_suppresses<MODE>(pl, pj, thr);
                    m = __ballot(sup);
                    pairs += (lane > l) && ((alive_mask >> lane) & 1ull) ? 1u : 0u;
                }
                if (lane == 0) s_sup[l] = m;
            }
        }
        __syncthreads();
        // wavefront sweep of the tile + emission of the kept rows
        if (wave == 0) {
            u64 alive = alive_mask, keptm = 0ull;
            int nk = nkept;
            for (int l = 0; l < 64; ++l) {
                if (((alive >> l) & 1ull) && nk < max_out) {
                    keptm |= 1ull << l;
                    ++nk;
                    alive &= ~s_sup[l];
                }
            }
            if (lane == 0) s_mask[1] = keptm;
            if ((keptm >> lane) & 1ull) {
                const int pos = __popcll(keptm & lanes_below(lane));
                const float4 cb4 = s_cbox[lane];
                const YpBox bl = {cb4.x, cb4.y, cb4.z, cb4.w};
                s_kprep[pos] = yp_prep<MODE>(bl);
                const int slot = s.sslot[reg + r0 + lane];
                const int m = nkept + pos;
                if (MODE == YP_MODE_REFERENCE) {
                    const size_t q = (size_t)b * s.max_out + m;
                    reinterpret_cast<float4 *>(s.out_boxes)[q] = cb4;
                    s.out_scores[q] = s.cand_key[ibase + slot];
                    s.out_classes[q] = s.cand_cls[ibase + slot];
                    const int tile = slot / ts;
                    s.out_index[q] = s.tile_prefix[(size_t)b * tpi + tile] + (slot - tile * ts);
                } else {
                    s.kept_slot[reg + m] = slot;
                }
            }
        }
        __syncthreads();
        const int nkc = __popcll(s_mask[1]);
        nkept += nkc;
        if (nkept >= max_out) break;
        // column pass: each thread tests the later columns it owns against the kept rows
        if (nkc > 0) {
            for (int j = r0 + 64 + tid; j < K; j += kSegThreads) {
                if (s.removed[reg + j]) continue;
                const float4 cj = s.sbox[reg + j];
                const YpBox bj = {cj.x, cj.y, cj.z, cj.w};
                const YpPrep pj = yp_prep<MODE>(bj);
                for (int q = 0; q < nkc; ++q) {
                    ++pairs;
                    if (yp_suppresses<MODE>(s_kprep[q], pj, thr)) {
                        s.removed[reg + j] = 1;
                        break;
                    }
                }
            }
        }
        __syncthreads();
    }

    pairs = (unsigned int)wave_sum((int)pairs);
    if (lane == 0 && pairs) atomicAdd(&s_pairs, pairs);
    __syncthreads();
    if (tid == 0) {
        s.seg_kept[seg] = nkept;
        s.seg_pairs[seg] = s_pairs;
        if (MODE == YP_MODE_REFERENCE) {
            s.out_counts[b] = nkept;
            s.surv_counts[b] = K;
        }
    }
}

// --------------------------------------------------------------------- final row gather
__global__ void k_gather_rows(GatherArgs g)
{
    __shared__ int s_part[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int seg = blockIdx.x;
    int part = 0;
    for (int i = tid; i < seg; i += 256) part += g.seg_kept[i];
    part = wave_sum(part);
    if (lane == 0) s_part[wave] = part;
    __syncthreads();
    const int prefix = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    const int nk = g.seg_kept[seg];
    if (tid == 0) {
        if (seg == g.n_seg - 1) g.out_count[0] = prefix + nk;
        if (g.stats) {
            g.stats[seg] = g.seg_cand[seg];
            g.stats[g.n_seg + seg] = nk;
            g.stats[2 * g.n_seg + seg] = (int)g.seg_pairs[seg];
        }
    }
    const int b = seg / g.C, c = seg - b * g.C;
    const size_t ibase = (size_t)b * g.spi;
    const size_t reg = ibase + g.seg_region[seg];
    for (int k = tid; k < nk; k += 256) {
        const long long d = (long long)prefix + k;
        if (d >= g.cap) break;
        const int slot = g.kept_slot[reg + k];
        const float4 bx = g.cand_box[ibase + slot];
        float4 *o = reinterpret_cast<float4 *>(g.out_rows + d * 8);
        o[0] = make_float4((float)(b + g.batch_offset), bx.x, bx.y, bx.z);
        o[1] = make_float4(bx.w, g.cand_key[ibase + slot], g.cand_aux[ibase + slot], (float)c);
    }
}

// ------------------------------------------------------------------- decode-only kernels
// YOLOv3 predict_transform for all heads: thread (wave = anchor, lane = cell) decodes one
// row into an LDS tile laid out exactly as the output, then the block streams the tile's
// contiguous 64*A*V floats out with coalesced stores.
__global__ void k_decode_heads_v3(HeadsArgs ha, float *out_rows)
{
    extern __shared__ float s_tile[];
    const int tile = blockIdx.x, b = blockIdx.y;
    int h = 0;
#pragma unroll
    for (int i = 1; i < YP_MAX_HEADS; ++i)
        if (i < ha.n_heads && tile >= ha.h[i].tile_begin) h = i;
    const HeadGeom &g = ha.h[h];
    const int lane = threadIdx.x & 63, a = threadIdx.x >> 6;
    const int C = ha.C, V = 5 + C;
    const int cells = g.gh * g.gw;
    const int cell0 = (tile - g.tile_begin) * 64;
    const int cell = cell0 + lane;
    if (a < g.A && cell < cells) {
        const float *p = g.data + ((size_t)(b * g.A + a) * V) * cells + cell;
        float *o = s_tile + (lane * g.A + a) * V;
        const int row = cell / g.gw, col = cell - row * g.gw;
        o[0] = (yp_sigmoid(p[0]) + (float)col) * g.sx;
        o[1] = (yp_sigmoid(p[(size_t)1 * cells]) + (float)row) * g.sy;
        o[2] = (yp_exp(p[(size_t)2 * cells]) * g.anc[2 * a + 0]) * g.sx;
        o[3] = (yp_exp(p[(size_t)3 * cells]) * g.anc[2 * a + 1]) * g.sy;
        o[4] = yp_sigmoid(p[(size_t)4 * cells]);
        for (int j = 0; j < C; ++j) o[5 + j] = yp_sigmoid(p[(size_t)(5 + j) * cells]);
    }
    __syncthreads();
    const int ncell = (cells - cell0) < 64 ? (cells - cell0) : 64;
    const int nflt = ncell * g.A * V;
    float *dst = out_rows + ((size_t)b * ha.N + g.row_begin + (size_t)cell0 * g.A) * V;
    for (int i = threadIdx.x; i < nflt; i += blockDim.x) dst[i] = s_tile[i];
}

// Reference predict_transform (detect_net.py:152-183): thread <-> row.
__global__ void k_decode_grid_ref(const float *pred, int H, int W, int A, int C, AnchorTable at,
                                  float *center, float *wh, float *obj, float *cls, long long total)
{
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= total) return;
    const int N = H * W * A, V = 5 + C;
    const int n = (int)(o % N);
    const int a = n % A, k = n / A;
    const float gx = (float)(k % H), gy = (float)(k / H);
    const float *t = pred + o * V;
    float2 ce, sz;
    ce.x = (yp_sigmoid(t[0]) + gx) / (float)H;
    ce.y = (yp_sigmoid(t[1]) + gy) / (float)W;
    sz.x = (yp_exp(t[2]) * at.anc[2 * a + 0]) / (float)H;
    sz.y = (yp_exp(t[3]) * at.anc[2 * a + 1]) / (float)W;
    reinterpret_cast<float2 *>(center)[o] = ce;
    reinterpret_cast<float2 *>(wh)[o] = sz;
    obj[o] = yp_sigmoid(t[4]);
    float m = t[5];
    for (int j = 1; j < C; ++j) m = t[5 + j] > m ? t[5 + j] : m;
    float s = 0.f;
    for (int j = 0; j < C; ++j) s = s + yp_exp(t[5 + j] - m);
    for (int j = 0; j < C; ++j) cls[o * C + j] = yp_exp(t[5 + j] - m) / s;
}

__global__ void k_bbox_iou(const float *box1, const float *boxes2, int m, int row_stride,
                           float *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const YpBox b1 = {box1[0], box1[1], box1[2], box1[3]};
    const float *q = boxes2 + (size_t)i * row_stride;
    const YpBox b2 = {q[0], q[1], q[2], q[3]};
    out[i] = yp_iou_v3(yp_prep_v3(b1), yp_prep_v3(b2));
}

// ------------------------------------------------------------------------------ host side

constexpr size_t kAlign = 256;

struct Workspace {
    float4 *cand_box;
    float *cand_key;
    float *cand_aux;
    int *cand_cls;
    int *tile_count;
    int *tile_prefix;
    u64 *gkeys;
    float4 *sbox;
    int *sslot;
    unsigned char *removed;
    int *kept_slot;
    int *seg_kept;
    int *seg_cand;
    int *seg_region;
    unsigned int *seg_pairs;
};

template <class T>
size_t carve_one(char *base, size_t off, size_t count, T **out)
{
    off = (off + kAlign - 1) / kAlign * kAlign;
    if (base) *out = reinterpret_cast<T *>(base + off);
    return off + count * sizeof(T);
}

// Lays the scratch arrays out inside `base` (or just sizes them when base is null).
size_t carve(void *base_v, size_t slots, size_t tiles, size_t n_seg, Workspace *w)
{
    char *base = static_cast<char *>(base_v);
    Workspace dummy;
    if (!w) w = &dummy;
    size_t off = 0;
    off = carve_one(base, off, slots, &w->cand_box);
    off = carve_one(base, off, slots, &w->cand_key);
    off = carve_one(base, off, slots, &w->cand_aux);
    off = carve_one(base, off, slots, &w->cand_cls);
    off = carve_one(base, off, tiles, &w->tile_count);
    off = carve_one(base, off, tiles, &w->tile_prefix);
    off = carve_one(base, off, 2 * slots, &w->gkeys);
    off = carve_one(base, off, slots, &w->sbox);
    off = carve_one(base, off, slots, &w->sslot);
    off = carve_one(base, off, slots, &w->removed);
    off = carve_one(base, off, slots, &w->kept_slot);
    off = carve_one(base, off, n_seg, &w->seg_kept);
    off = carve_one(base, off, n_seg, &w->seg_cand);
    off = carve_one(base, off, n_seg, &w->seg_region);
    off = carve_one(base, off, n_seg, &w->seg_pairs);
    return (off + kAlign - 1) / kAlign * kAlign;
}

bool params_ok(const YpParams *p, int mode)
{
    if (!p || p->mode != mode || p->batch <= 0 || p->batch > 65535) return false;
    if (p->num_classes <= 0 || p->num_classes > 4096) return false;
    if (!(p->nms_thr == p->nms_thr) || !(p->conf_thr == p->conf_thr)) return false;
    return true;
}

// Geometry of a set of YOLOv3 heads: tiles of 64 cells, slots per tile = 64 * max anchors.
int heads_geometry(const YpHead *heads, int n_heads, int C, HeadsArgs *ha, int *amax_out)
{
    if (!heads || n_heads < 1 || n_heads > YP_MAX_HEADS) return YP_ERR_ARG;
    int amax = 0;
    long long tiles = 0, rows = 0;
    for (int i = 0; i < n_heads; ++i) {
        const YpHead &h = heads[i];
        if (!h.data || h.grid_h <= 0 || h.grid_w <= 0) return YP_ERR_ARG;
        if (h.num_anchors < 1 || h.num_anchors > YP_MAX_ANCHORS) return YP_ERR_UNSUPPORTED;
        if ((long long)h.grid_h * h.grid_w > (1ll << 24)) return YP_ERR_UNSUPPORTED;
        HeadGeom &g = ha->h[i];
        g.data = h.data;
        g.gh = h.grid_h;
        g.gw = h.grid_w;
        g.A = h.num_anchors;
        g.sx = h.stride_x;
        g.sy = h.stride_y;
        g.tile_begin = (int)tiles;
        g.row_begin = (int)rows;
        for (int k = 0; k < 2 * YP_MAX_ANCHORS; ++k) g.anc[k] = h.anchors[k];
        tiles += ((long long)h.grid_h * h.grid_w + 63) / 64;
        rows += (long long)h.grid_h * h.grid_w * h.num_anchors;
        amax = h.num_anchors > amax ? h.num_anchors : amax;
    }
    for (int i = n_heads; i < YP_MAX_HEADS; ++i) ha->h[i] = ha->h[0];
    if (tiles > (1ll << 22) || rows > (1ll << 28)) return YP_ERR_UNSUPPORTED;
    ha->n_heads = n_heads;
    ha->tpi = (int)tiles;
    ha->C = C;
    ha->N = (int)rows;
    *amax_out = amax;
    return YP_OK;
}

bool slots_fit(long long batch, long long spi)
{
    return batch * spi > 0 && batch * spi < (1ll << 31);
}

bool workspace_ok(const void *ws, size_t have, size_t need)
{
    return ws && (reinterpret_cast<uintptr_t>(ws) % kAlign) == 0 && have >= need;
}

SegArgs make_seg_args(const Workspace &w, const YpParams *p, int spi, int tpi, int ts,
                      int per_class)
{
    SegArgs s = {};
    s.B = p->batch;
    s.C = p->num_classes;
    s.spi = spi;
    s.tpi = tpi;
    s.ts = ts;
    s.per_class = per_class;
    s.max_out = p->max_out;
    s.batch_offset = p->batch_offset;
    s.thr = p->nms_thr;
    s.cand_box = w.cand_box;
    s.cand_key = w.cand_key;
    s.cand_aux = w.cand_aux;
    s.cand_cls = w.cand_cls;
    s.tile_count = w.tile_count;
    s.tile_prefix = w.tile_prefix;
    s.gkeys = w.gkeys;
    s.sbox = w.sbox;
    s.sslot = w.sslot;
    s.removed = w.removed;
    s.kept_slot = w.kept_slot;
    s.seg_kept = w.seg_kept;
    s.seg_cand = w.seg_cand;
    s.seg_region = w.seg_region;
    s.seg_pairs = w.seg_pairs;
    return s;
}

// Optional per-kernel timing: when enabled, every detect call records four HIP events
// (before the front kernel and after each of its three launches) on the call's stream.
struct ProfileState {
    bool on = false;
    int max_calls = 0;
    int calls = 0;
    hipEvent_t *ev = nullptr;  // 4 per call
};
ProfileState g_prof;

inline void prof_mark(hipStream_t st, int k)
{
    if (g_prof.on && g_prof.calls < g_prof.max_calls) (void)hipEventRecord(g_prof.ev[4 * g_prof.calls + k], st);
}

inline void prof_next()
{
    if (g_prof.on && g_prof.calls < g_prof.max_calls) ++g_prof.calls;
}

// Segment NMS + final gather of the YOLOv3 path (shared by the heads and rows entries).
int finish_v3(const Workspace &w, const YpParams *p, int spi, int tpi, int ts, float *out_rows,
              long long cap, int32_t *out_count, int32_t *stats, hipStream_t st)
{
    const int n_seg = p->batch * p->num_classes;
    SegArgs s = make_seg_args(w, p, spi, tpi, ts, 1);
    s.max_out = p->max_out;
    hipLaunchKernelGGL(k_segment_nms<YP_MODE_YOLOV3>, dim3(n_seg), dim3(kSegThreads), 0, st, s);
    prof_mark(st, 2);
    GatherArgs g = {};
    g.n_seg = n_seg;
    g.C = p->num_classes;
    g.spi = spi;
    g.batch_offset = p->batch_offset;
    g.cand_box = w.cand_box;
    g.cand_key = w.cand_key;
    g.cand_aux = w.cand_aux;
    g.kept_slot = w.kept_slot;
    g.seg_kept = w.seg_kept;
    g.seg_cand = w.seg_cand;
    g.seg_region = w.seg_region;
    g.seg_pairs = w.seg_pairs;
    g.out_rows = out_rows;
    g.cap = cap;
    g.out_count = out_count;
    g.stats = stats;
    hipLaunchKernelGGL(k_gather_rows, dim3(n_seg), dim3(256), 0, st, g);
    prof_mark(st, 3);
    prof_next();
    return (int)hipGetLastError();
}

int finish_ref(const Workspace &w, const YpParams *p, int spi, int tpi, int ts, float *out_boxes,
               float *out_scores, int32_t *out_classes, int32_t *out_index, int32_t *out_counts,
               int32_t *surv_counts, hipStream_t st)
{
    SegArgs s = make_seg_args(w, p, spi, tpi, ts, 0);
    s.out_boxes = out_boxes;
    s.out_scores = out_scores;
    s.out_classes = out_classes;
    s.out_index = out_index;
    s.out_counts = out_counts;
    s.surv_counts = surv_counts;
    hipLaunchKernelGGL(k_segment_nms<YP_MODE_REFERENCE>, dim3(p->batch), dim3(kSegThreads), 0, st, s);
    prof_mark(st, 2);
    prof_mark(st, 3);
    prof_next();
    return (int)hipGetLastError();
}

}  // namespace

// ================================================================================= C ABI

extern "C" {

int yp_version(void) { return 100; }

const char *yp_strerror(int code)
{
    switch (code) {
    case YP_OK: return "ok";
    case YP_ERR_ARG: return "invalid argument";
    case YP_ERR_WORKSPACE: return "workspace missing, misaligned or too small";
    case YP_ERR_UNSUPPORTED: return "shape beyond the compiled limits";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

int yp_predict_transform_reference(const float *d_pred, int batch, int grid_h, int grid_w,
                                   int num_anchors, int num_classes, const float *anchors,
                                   float *d_center, float *d_wh, float *d_obj, float *d_cls,
                                   void *stream)
{
    if (!d_pred || !anchors || !d_center || !d_wh || !d_obj || !d_cls) return YP_ERR_ARG;
    if (batch <= 0 || grid_h <= 0 || grid_w <= 0 || num_classes <= 0) return YP_ERR_ARG;
    if (num_anchors < 1 || num_anchors > YP_MAX_ANCHORS) return YP_ERR_UNSUPPORTED;
    AnchorTable at = {};
    for (int i = 0; i < 2 * num_anchors; ++i) at.anc[i] = anchors[i];
    const long long total = (long long)batch * grid_h * grid_w * num_anchors;
    if (total >= (1ll << 31)) return YP_ERR_UNSUPPORTED;
    const int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(k_decode_grid_ref, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_pred,
                       grid_h, grid_w, num_anchors, num_classes, at, d_center, d_wh, d_obj, d_cls,
                       total);
    return (int)hipGetLastError();
}

int yp_predict_transform_yolov3(const YpHead *heads, int n_heads, int batch, int num_classes,
                                float *d_rows, void *stream)
{
    if (!d_rows || batch <= 0 || batch > 65535 || num_classes <= 0 || num_classes > 4096)
        return YP_ERR_ARG;
    HeadsArgs ha;
    int amax = 0;
    const int rc = heads_geometry(heads, n_heads, num_classes, &ha, &amax);
    if (rc != YP_OK) return rc;
    const size_t lds = (size_t)64 * amax * (5 + num_classes) * sizeof(float);
    if (lds > 64 * 1024) return YP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_decode_heads_v3, dim3(ha.tpi, batch), dim3(64 * amax), lds,
                       (hipStream_t)stream, ha, d_rows);
    return (int)hipGetLastError();
}

size_t yp_detect_yolov3_workspace_bytes(const YpHead *heads, int n_heads, int batch,
                                        int num_classes)
{
    HeadsArgs ha;
    int amax = 0;
    YpHead tmp[YP_MAX_HEADS];
    if (!heads || n_heads < 1 || n_heads > YP_MAX_HEADS || batch <= 0 || num_classes <= 0) return 0;
    static const float dummy = 0.f;
    for (int i = 0; i < n_heads; ++i) {
        tmp[i] = heads[i];
        if (!tmp[i].data) tmp[i].data = &dummy;  // sizing does not need the data pointers
    }
    if (heads_geometry(tmp, n_heads, num_classes, &ha, &amax) != YP_OK) return 0;
    const long long spi = (long long)ha.tpi * 64 * amax;
    if (!slots_fit(batch, spi)) return 0;
    return carve(nullptr, (size_t)batch * spi, (size_t)batch * ha.tpi,
                 (size_t)batch * num_classes, nullptr);
}

int yp_detect_yolov3(const YpHead *heads, int n_heads, const YpParams *params, float *d_out_rows,
                     int64_t out_cap, int32_t *d_out_count, int32_t *d_stats, void *d_workspace,
                     size_t workspace_bytes, void *stream)
{
    if (!params_ok(params, YP_MODE_YOLOV3) || !d_out_rows || !d_out_count || out_cap < 0)
        return YP_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(d_out_rows) & 15) != 0) return YP_ERR_ARG;
    HeadsArgs ha;
    int amax = 0;
    const int rc = heads_geometry(heads, n_heads, params->num_classes, &ha, &amax);
    if (rc != YP_OK) return rc;
    const int ts = 64 * amax;
    const long long spi = (long long)ha.tpi * ts;
    if (!slots_fit(params->batch, spi)) return YP_ERR_UNSUPPORTED;
    const size_t slots = (size_t)params->batch * spi, tiles = (size_t)params->batch * ha.tpi;
    const size_t n_seg = (size_t)params->batch * params->num_classes;
    Workspace w;
    const size_t need = carve(nullptr, slots, tiles, n_seg, nullptr);
    if (!workspace_ok(d_workspace, workspace_bytes, need)) return YP_ERR_WORKSPACE;
    carve(d_workspace, slots, tiles, n_seg, &w);
    CandBuf cb = {w.cand_box, w.cand_key, w.cand_aux, w.cand_cls, w.tile_count, ha.tpi, ts};
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(ha.tpi, params->batch), block(64 * amax);
    prof_mark(st, 0);
    if (params->num_classes == 5)
        hipLaunchKernelGGL(k_front_heads_v3<5>, grid, block, 0, st, ha, cb, params->conf_thr);
    else
        hipLaunchKernelGGL(k_front_heads_v3<0>, grid, block, 0, st, ha, cb, params->conf_thr);
    prof_mark(st, 1);
    return finish_v3(w, params, (int)spi, ha.tpi, ts, d_out_rows, out_cap, d_out_count, d_stats, st);
}

size_t yp_rows_workspace_bytes(int batch, int rows_per_image, int segments_per_image)
{
    if (batch <= 0 || rows_per_image <= 0 || segments_per_image <= 0) return 0;
    const long long tpi = ((long long)rows_per_image + kRowTile - 1) / kRowTile;
    if (!slots_fit(batch, tpi * kRowTile)) return 0;
    return carve(nullptr, (size_t)batch * tpi * kRowTile, (size_t)batch * tpi,
                 (size_t)batch * segments_per_image, nullptr);
}

size_t yp_detect_reference_workspace_bytes(int batch, int grid_h, int grid_w, int num_anchors)
{
    if (grid_h <= 0 || grid_w <= 0 || num_anchors <= 0) return 0;
    const long long n = (long long)grid_h * grid_w * num_anchors;
    if (n >= (1ll << 28)) return 0;
    return yp_rows_workspace_bytes(batch, (int)n, 1);
}

int yp_write_results(const float *d_rows, int rows_per_image, const YpParams *params,
                     float *d_out_rows, int64_t out_cap, int32_t *d_out_count, int32_t *d_stats,
                     void *d_workspace, size_t workspace_bytes, void *stream)
{
    if (!params_ok(params, YP_MODE_YOLOV3) || !d_rows || !d_out_rows || !d_out_count ||
        out_cap < 0 || rows_per_image <= 0)
        return YP_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(d_out_rows) & 15) != 0) return YP_ERR_ARG;
    const int tpi = (rows_per_image + kRowTile - 1) / kRowTile;
    const long long spi = (long long)tpi * kRowTile;
    if (!slots_fit(params->batch, spi)) return YP_ERR_UNSUPPORTED;
    const size_t slots = (size_t)params->batch * spi, tiles = (size_t)params->batch * tpi;
    Workspace w;
    const size_t need = yp_rows_workspace_bytes(params->batch, rows_per_image, params->num_classes);
    if (!workspace_ok(d_workspace, workspace_bytes, need)) return YP_ERR_WORKSPACE;
    carve(d_workspace, slots, tiles, (size_t)params->batch * params->num_classes, &w);
    CandBuf cb = {w.cand_box, w.cand_key, w.cand_aux, w.cand_cls, w.tile_count, tpi, kRowTile};
    hipStream_t st = (hipStream_t)stream;
    prof_mark(st, 0);
    hipLaunchKernelGGL(k_front_rows_v3, dim3(tpi, params->batch), dim3(kRowTile), 0, st, d_rows,
                       rows_per_image, params->num_classes, cb, params->conf_thr);
    prof_mark(st, 1);
    return finish_v3(w, params, (int)spi, tpi, kRowTile, d_out_rows, out_cap, d_out_count, d_stats,
                     st);
}

static int ref_outputs_ok(const YpParams *p, const void *a, const void *b, const void *c,
                          const void *d, const void *e, const void *f)
{
    if (!params_ok(p, YP_MODE_REFERENCE) || !a || !b || !c || !d || !e || !f) return 0;
    if (p->max_out <= 0 || p->max_out > (1 << 20)) return 0;
    if ((reinterpret_cast<uintptr_t>(a) & 15) != 0) return 0;
    return 1;
}

int yp_detect_reference(const YpHead *head, const YpParams *params, float *d_out_boxes,
                        float *d_out_scores, int32_t *d_out_classes, int32_t *d_out_index,
                        int32_t *d_out_counts, int32_t *d_surv_counts, void *d_workspace,
                        size_t workspace_bytes, void *stream)
{
    if (!head || !head->data) return YP_ERR_ARG;
    if (!ref_outputs_ok(params, d_out_boxes, d_out_scores, d_out_classes, d_out_index,
                        d_out_counts, d_surv_counts))
        return YP_ERR_ARG;
    if (head->grid_h <= 0 || head->grid_w <= 0) return YP_ERR_ARG;
    if (head->num_anchors < 1 || head->num_anchors > YP_MAX_ANCHORS) return YP_ERR_UNSUPPORTED;
    const long long n = (long long)head->grid_h * head->grid_w * head->num_anchors;
    if (n >= (1ll << 28)) return YP_ERR_UNSUPPORTED;
    const int tpi = (int)((n + kRowTile - 1) / kRowTile);
    const long long spi = (long long)tpi * kRowTile;
    if (!slots_fit(params->batch, spi)) return YP_ERR_UNSUPPORTED;
    const size_t need = yp_rows_workspace_bytes(params->batch, (int)n, 1);
    if (!workspace_ok(d_workspace, workspace_bytes, need)) return YP_ERR_WORKSPACE;
    Workspace w;
    carve(d_workspace, (size_t)params->batch * spi, (size_t)params->batch * tpi,
          (size_t)params->batch, &w);
    CandBuf cb = {w.cand_box, w.cand_key, w.cand_aux, w.cand_cls, w.tile_count, tpi, kRowTile};
    AnchorTable at = {};
    for (int i = 0; i < 2 * head->num_anchors; ++i) at.anc[i] = head->anchors[i];
    hipStream_t st = (hipStream_t)stream;
    prof_mark(st, 0);
    hipLaunchKernelGGL(k_front_grid_ref, dim3(tpi, params->batch), dim3(kRowTile), 0, st,
                       head->data, head->grid_h, head->grid_w, head->num_anchors,
                       params->num_classes, at, cb, params->conf_thr, params->im_h, params->im_w);
    prof_mark(st, 1);
    return finish_ref(w, params, (int)spi, tpi, kRowTile, d_out_boxes, d_out_scores, d_out_classes,
                      d_out_index, d_out_counts, d_surv_counts, st);
}

int yp_filter_boxes_reference(const float *d_center, const float *d_wh, const float *d_obj,
                              const float *d_cls, int rows_per_image, const YpParams *params,
                              float *d_out_boxes, float *d_out_scores, int32_t *d_out_classes,
                              int32_t *d_out_index, int32_t *d_out_counts,
                              int32_t *d_surv_counts, void *d_workspace, size_t workspace_bytes,
                              void *stream)
{
    if (!d_center || !d_wh || !d_obj || !d_cls || rows_per_image <= 0) return YP_ERR_ARG;
    if (!ref_outputs_ok(params, d_out_boxes, d_out_scores, d_out_classes, d_out_index,
                        d_out_counts, d_surv_counts))
        return YP_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(d_center) & 7) || (reinterpret_cast<uintptr_t>(d_wh) & 7))
        return YP_ERR_ARG;
    const int tpi = (rows_per_image + kRowTile - 1) / kRowTile;
    const long long spi = (long long)tpi * kRowTile;
    if (!slots_fit(params->batch, spi)) return YP_ERR_UNSUPPORTED;
    const size_t need = yp_rows_workspace_bytes(params->batch, rows_per_image, 1);
    if (!workspace_ok(d_workspace, workspace_bytes, need)) return YP_ERR_WORKSPACE;
    Workspace w;
    carve(d_workspace, (size_t)params->batch * spi, (size_t)params->batch * tpi,
          (size_t)params->batch, &w);
    CandBuf cb = {w.cand_box, w.cand_key, w.cand_aux, w.cand_cls, w.tile_count, tpi, kRowTile};
    hipStream_t st = (hipStream_t)stream;
    prof_mark(st, 0);
    hipLaunchKernelGGL(k_front_split_ref, dim3(tpi, params->batch), dim3(kRowTile), 0, st, d_center,
                       d_wh, d_obj, d_cls, rows_per_image, params->num_classes, cb,
                       params->conf_thr, params->im_h, params->im_w);
    prof_mark(st, 1);
    return finish_ref(w, params, (int)spi, tpi, kRowTile, d_out_boxes, d_out_scores, d_out_classes,
                      d_out_index, d_out_counts, d_surv_counts, st);
}

int yp_profile_start(int max_calls)
{
    if (max_calls <= 0 || max_calls > (1 << 16) || g_prof.on) return YP_ERR_ARG;
    g_prof.ev = new hipEvent_t[4 * (size_t)max_calls];
    for (int i = 0; i < 4 * max_calls; ++i) {
        const hipError_t e = hipEventCreate(&g_prof.ev[i]);
        if (e != hipSuccess) return (int)e;
    }
    g_prof.max_calls = max_calls;
    g_prof.calls = 0;
    g_prof.on = true;
    return YP_OK;
}

int yp_profile_stop(double *front_ms, double *nms_ms, double *gather_ms, int *calls)
{
    if (!g_prof.on || !front_ms || !nms_ms || !gather_ms || !calls) return YP_ERR_ARG;
    g_prof.on = false;
    double acc[3] = {0.0, 0.0, 0.0};
    int rc = YP_OK;
    if (g_prof.calls > 0) {
        const hipError_t e = hipEventSynchronize(g_prof.ev[4 * (g_prof.calls - 1) + 3]);
        if (e != hipSuccess) rc = (int)e;
    }
    for (int c = 0; c < g_prof.calls && rc == YP_OK; ++c)
        for (int k = 0; k < 3; ++k) {
            float ms = 0.f;
            const hipError_t e = hipEventElapsedTime(&ms, g_prof.ev[4 * c + k], g_prof.ev[4 * c + k + 1]);
            if (e != hipSuccess) { rc = (int)e; break; }
            acc[k] += ms;
        }
    *calls = g_prof.calls;
    const double n = g_prof.calls > 0 ? (double)g_prof.calls : 1.0;
    *front_ms = acc[0] / n;
    *nms_ms = acc[1] / n;
    *gather_ms = acc[2] / n;
    for (int i = 0; i < 4 * g_prof.max_calls; ++i) (void)hipEventDestroy(g_prof.ev[i]);
    delete[] g_prof.ev;
    g_prof.ev = nullptr;
    g_prof.max_calls = g_prof.calls = 0;
    return rc;
}

int yp_bbox_iou(const float *d_box1, const float *d_boxes2, int m, int row_stride, float *d_out,
                void *stream)
{
    if (!d_box1 || !d_boxes2 || !d_out || m < 0 || row_stride < 4) return YP_ERR_ARG;
    if (m == 0) return YP_OK;
    hipLaunchKernelGGL(k_bbox_iou, dim3((m + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_box1,
                       d_boxes2, m, row_stride, d_out);
    return (int)hipGetLastError();
}

}  // extern "C"
