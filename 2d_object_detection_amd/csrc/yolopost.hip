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
#include <hip/hip_ext.h>

#include <cstdlib>
#include <mutex>

#include "yolopost.h"
#include "yp_math.h"
#include "yp_tile.h"

namespace {

typedef unsigned long long u64;

constexpr int kRowTile = 256;     // rows per tile in the row-mapped front kernels
constexpr int kFrontGroups = 2;   // 64-cell groups per tile of the YOLOv3 heads front kernel
constexpr int kFrontCells = 64 * kFrontGroups;
constexpr int kStageWin = 128;    // survivors staged per round by the single-wave front kernel
constexpr int kSegThreads = 512;  // threads per NMS workgroup (the kernel is only validated for 512)
constexpr int kSegWaves = kSegThreads / 64;
constexpr int kSortCap = 2048;    // keys the first part of the sort block (s_raw) holds
constexpr int kBigSortCap = 4096; // largest segment gathered into and sorted in LDS (the whole block)
constexpr int kSmallCap = 512;    // segments up to this size use the all-pairs LDS bitmask path
constexpr int kSmallWords = kSmallCap / 64;
// words of the triangular pairwise bitmask of a kSmallCap segment: sum_w (cap - 64 w)
constexpr int kMaskWords = kSmallWords * kSmallCap - 32 * kSmallWords * (kSmallWords - 1);
constexpr int kRawBytes = (kSortCap * 8 > kMaskWords * 8) ? kSortCap * 8 : kMaskWords * 8;
constexpr int kSmallBytes = 32 * (kSmallCap + 8) + 16 * 32;   // + 32 boxes the pair loop may over-read
constexpr int kGraphMinCand = 768;      // segments with more candidates can take the graph path
constexpr int kGraphPairGrid = 1024;    // workgroups of k_graph_pairs at most: what is resident at 4 waves per SIMD
struct GraphSeg;

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
    HeadGeom h[YP_MAX_HEADS];   // geometry shared by every set; h[i].data = head i of set 0
    // Head tensors of every set (independent batches of identical geometry handled by one
    // launch): set s, head i.  blockIdx.z selects the set; image b of set s is image
    // s * bps + b of the launch, which is how the workspace and the segments number it.
    const float *data[YP_MAX_SETS][YP_MAX_HEADS];
    int n_heads;
    int tpi;   // tiles per image
    int C;     // classes
    int N;     // rows per image over all heads
    int bps;   // images per set
    unsigned long long *dbg;  // diagnostic builds only (YP_STAMPS)
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
    int *ctrl;  // launch control words, cleared by the front kernel: [0] some segment took the
                // graph path, [1] bin tables handed out, [2] batches of 64 boxes in those segments
};

// The first thread of a front kernel clears the control words of the launch (the kernels that
// follow on the stream set and read them).
__device__ __forceinline__ void clear_ctrl(const CandBuf &cb)
{
    if (cb.ctrl && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
        cb.ctrl[0] = 0;
        cb.ctrl[1] = 0;
        cb.ctrl[2] = 0;
    }
}

struct SegArgs {
    int B, C, spi, tpi, ts, per_class, max_out, batch_offset;
    float thr;
    const float4 *cand_box;
    const float *cand_key;
    const float *cand_aux;
    const int *cand_cls;
    const int *tile_count;
    int *tile_prefix;        // [B*tpi] exclusive prefix of tile_count inside an image
    u64 *gkeys;              // [2*B*spi] sort area for segments larger than kBigSortCap, then their
                             // boxes in x order
    float4 *sbox;            // [B*spi] boxes in sorted order, per segment region
    int *sslot;              // [B*spi] slot ids in sorted order
    unsigned char *removed;  // [B*spi] suppression flags in sorted order
    int *kept_slot;          // [B*spi] slot ids of kept rows, per segment region
    int *xidx;               // [B*spi] large segments: sorted positions of the live columns in x order
    int *seg_kept;           // [n_seg]
    int *seg_cand;           // [n_seg]
    int *seg_region;         // [n_seg] region start inside the image
    unsigned int *seg_pairs; // [n_seg] IoU evaluations (low 32 bits)
    unsigned int *seg_ties;  // [n_seg] evaluated pairs whose IoU lay within 2^-20 (relative) of the
                             // threshold, i.e. that needed the exact divide-and-compare
    GraphSeg *gseg;          // [n_seg] index descriptors of the graph path (null: path disabled)
    int *bin_tables;         // pool of bin offset tables for it
    int *glist;              // graph segments of the launch, indexed like the tables
    int *gcnt, *gedge;       // suppressor records the graph kernels fill (cleared here)
    int *ctrl;               // launch control words (see CandBuf)
    int *img_flag;           // [B] set by k_image_nms: 0 = that kernel finished the image, 1 = it is
                             // left to k_segment_nms (null: k_image_nms did not run)
    // reference-mode outputs, padded to max_out rows per image
    float *out_boxes;
    float *out_scores;
    int *out_classes;
    int *out_index;
    int *out_counts;
    int *surv_counts;
    unsigned long long *dbg;  // diagnostic builds only (YP_STAMPS): 32 time stamps per segment
};

// In-kernel time stamps exist only in the separate diagnostic build (-DYP_STAMPS); the
// shipped library compiles STAMP to nothing.
#ifdef YP_STAMPS
#define STAMP(k)                                                                        \
    do {                                                                                \
        if (threadIdx.x == 0 && s.dbg) {                                                \
            s.dbg[(size_t)stamp_block * 32 + (k)] = wall_clock64();                  \
            s.dbg[(size_t)stamp_block * 32 + 16 + (k)] = clock64();                  \
        }                                                                               \
    } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

struct GatherArgs {
    int n_seg, C, spi, batch_offset;
    int spset;               // segments per set (images per set x classes)
    long long set_stride;    // floats between the output rows of consecutive sets
    // YP_FLAG_RESCALE: kept boxes leave the network-input frame for the original one
    // (x - pad_x) / scale clamped to [0, im_w], likewise y; NMS has run on the unscaled boxes
    int rescale;
    float pad_x, pad_y, scale, im_w, im_h;
    const float4 *cand_box;
    const float *cand_key;
    const float *cand_aux;
    const int *kept_slot;
    const int *seg_kept;
    const int *seg_cand;
    const int *seg_region;
    const unsigned int *seg_pairs;
    const unsigned int *seg_ties;
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

// Single-instruction max/min.  For non-NaN operands they equal (a > b ? a : b) and
// (a < b ? a : b); written as asm because the compiler pairs every such select with a
// compare, and canonicalises the operands of the fmaxf/fminf/fmed3 builtins first.
// (The CPU emulation the tests run these kernels under defines the two macros itself.)
#ifndef YP_VMAX_F32
#define YP_VMAX_F32(r, a, b) asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b))
#define YP_VMIN_F32(r, a, b) asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b))
#endif
__device__ __forceinline__ float vmax(float a, float b)
{
    float r;
    YP_VMAX_F32(r, a, b);
    return r;
}
__device__ __forceinline__ float vmin(float a, float b)
{
    float r;
    YP_VMIN_F32(r, a, b);
    return r;
}

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

// Ordered rank inside a heads tile.  The block's waves are (group m, anchor a) pairs:
// wave = m * n_anchor + a holds anchor a of the 64 cells of group m, and the output row
// order is cell-major, anchor-minor (row = cell * A + anchor).  So a survivor is preceded
// by every survivor of earlier groups, by the survivors of earlier cells of its group (any
// anchor) and by the earlier anchors of its own cell.
__device__ __forceinline__ int tile_rank_cells(bool pass, u64 *s_bal, int n_anchor, int m, int a,
                                               int &total)
{
    const int lane = threadIdx.x & 63;
    const u64 bal = __ballot(pass);
    if (lane == 0) s_bal[m * n_anchor + a] = bal;
    __syncthreads();
    const u64 below = lanes_below(lane);
    int rank = 0;
    total = 0;
#pragma unroll
    for (int wm = 0; wm < kFrontGroups; ++wm) {
        for (int wa = 0; wa < n_anchor; ++wa) {
            const u64 mk = s_bal[wm * n_anchor + wa];
            const int c = __popcll(mk);
            total += c;
            if (wm < m) rank += c;
            else if (wm == m) {
                rank += __popcll(mk & below);
                if (wa < a) rank += (int)((mk >> lane) & 1ull);
            }
        }
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
// Block = kFrontCells cells x A anchors of one head of one image; wave (m, a) handles
// anchor a of cell group m, so every channel load is 64 consecutive floats of one NCHW
// plane (256 B per wave instruction).
//
// Phase 1 (all lanes): load the channels, decide the confidence filter.  sigmoid is
// monotone, so objectness logits outside the guard band [t_lo, t_hi] around logit(conf)
// are decided by a compare; only logits inside the band evaluate the sigmoid (the host
// passes an infinite band when conf is too close to 0 or 1 for the band to be safe).
// Survivors are ranked in anchor order with ballots and park their raw logits in LDS.
// Phase 2 (dense): thread r decodes the tile's r-th survivor, so the transcendental work is
// done for survivors only instead of by every wave that contains one.
template <int CS>
__global__ void k_front_heads_v3(HeadsArgs ha, CandBuf cb, float conf, float t_lo, float t_hi)
{
    __shared__ u64 s_bal[kFrontGroups * YP_MAX_ANCHORS];
    extern __shared__ float s_stage[];
    constexpr int kStride = 6 + CS;  // t0..t4, packed (anchor, cell), CS class logits
    const int tile = blockIdx.x, bl = blockIdx.y, set = blockIdx.z;
    const int b = set * ha.bps + bl;     // image index inside the launch
    int h = 0;
#pragma unroll
    for (int i = 1; i < YP_MAX_HEADS; ++i)
        if (i < ha.n_heads && tile >= ha.h[i].tile_begin) h = i;
    const HeadGeom &g = ha.h[h];
    const float *gdata = ha.data[set][h];
    const int n_anchor = blockDim.x / kFrontCells;
    // wave-uniform quantities are made scalar so that the plane base addresses are computed
    // once on the scalar unit and every load is "scalar base + lane offset"
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = wave / n_anchor, a = wave - m * n_anchor;
    const int C = CS ? CS : ha.C;
    const int V = 5 + C;
    const int cells = g.gh * g.gw;
    const int cell = (tile - g.tile_begin) * kFrontCells + m * 64 + lane;
    const bool active = (a < g.A) && (cell < cells);

    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f, t4 = 0.f;
    float tc[CS ? CS : 1];
    if (active) {
        const float *plane = gdata + ((size_t)(bl * g.A + a) * V) * cells;   // scalar
        t4 = (plane + (size_t)4 * cells)[cell];
        t0 = plane[cell];
        t1 = (plane + (size_t)1 * cells)[cell];
        t2 = (plane + (size_t)2 * cells)[cell];
        t3 = (plane + (size_t)3 * cells)[cell];
        if (CS) {
#pragma unroll
            for (int j = 0; j < (CS ? CS : 1); ++j) tc[j] = (plane + (size_t)(5 + j) * cells)[cell];
        }
    }
    bool pass = active && (t4 >= t_hi);
    if (active && !pass && !(t4 <= t_lo)) {
        const float obj = yp_sigmoid(t4);
        pass = (obj > conf) && (obj != 0.0f);
    }
    int total;
    const int rank = tile_rank_cells(pass, s_bal, n_anchor, m, a, total);
    if (pass) {
        float *st = s_stage + rank * kStride;
        st[0] = t0; st[1] = t1; st[2] = t2; st[3] = t3; st[4] = t4;
        reinterpret_cast<int *>(st)[5] = (a << 24) | cell;
        if (CS) {
#pragma unroll
            for (int j = 0; j < (CS ? CS : 1); ++j) st[6 + j] = tc[j];
        }
    }
    __syncthreads();
    for (int r = threadIdx.x; r < total; r += blockDim.x) {
        const float *st = s_stage + r * kStride;
        const int packed = reinterpret_cast<const int *>(st)[5];
        const int sa = packed >> 24, scell = packed & 0xffffff;
        const int row = scell / g.gw, col = scell - row * g.gw;
        const float obj = yp_sigmoid(st[4]);
        const float cx = (yp_sigmoid(st[0]) + (float)col) * g.sx;
        const float cy = (yp_sigmoid(st[1]) + (float)row) * g.sy;
        const float w = (yp_exp(st[2]) * g.anc[2 * sa + 0]) * g.sx;
        const float hgt = (yp_exp(st[3]) * g.anc[2 * sa + 1]) * g.sy;
        float best;
        int bi = 0;
        if (CS) {
            best = yp_sigmoid(st[6]);
#pragma unroll
            for (int j = 1; j < (CS ? CS : 1); ++j) {
                const float sc = yp_sigmoid(st[6 + j]);
                if (sc > best) { best = sc; bi = j; }
            }
        } else {
            const float *p = gdata + ((size_t)(bl * g.A + sa) * V) * cells + scell;
            best = yp_sigmoid(p[(size_t)5 * cells]);
            for (int j = 1; j < C; ++j) {
                const float sc = yp_sigmoid(p[(size_t)(5 + j) * cells]);
                if (sc > best) { best = sc; bi = j; }
            }
        }
        const float hw = w * 0.5f, hh = hgt * 0.5f;
        YpBox bx = {cx - hw, cy - hh, cx + hw, cy + hh};
        store_cand(cb, ((size_t)b * cb.tpi + tile) * cb.ts + r, bx, obj + 0.0f, best, bi);
    }
    if (threadIdx.x == 0) cb.tile_count[(size_t)b * cb.tpi + tile] = total;
    clear_ctrl(cb);
}

// One-wave variant of the heads front kernel for a compile-time class count CS and anchor
// count NA: the workgroup is a single wave that owns the tile's 128 cells x NA anchors, two
// CONSECUTIVE cells per lane, so a channel load is one 8-byte access per lane (512 B per
// wave instruction) whenever the plane slice is 8-byte aligned (always for even grids).  All
// 2*NA*(5+CS) values of a lane are requested before anything is consumed, the ballots that
// rank the survivors are wave-local, and the LDS staging between the two phases is written
// and read by the same wave -- so the kernel has no barrier at all and a quarter of the
// waves of the generic kernel above.  Same slots, same results.
template <int CS, int NA>
__global__ __launch_bounds__(64)
void k_front_heads_v3_wave(HeadsArgs ha, CandBuf cb, float conf,
                                                            float t_lo, float t_hi)
{
    extern __shared__ __align__(16) float s_stage[];
    constexpr int kStride = (6 + CS + 3) & ~3;   // t0..t4, packed (anchor, cell), CS class logits;
                                                 // padded so that a record is whole float4s
    constexpr int V = 5 + CS;
    static_assert(kFrontCells == 128, "two cells per lane");
    const int tile = blockIdx.x, bl = blockIdx.y, set = blockIdx.z;
    const int b = set * ha.bps + bl;     // image index inside the launch
    int h = 0;
#pragma unroll
    for (int i = 1; i < YP_MAX_HEADS; ++i)
        if (i < ha.n_heads && tile >= ha.h[i].tile_begin) h = i;
    const HeadGeom &g = ha.h[h];
    const float *gdata = ha.data[set][h];
    const int lane = threadIdx.x;
#ifdef YP_STAMPS
    const size_t dbg_blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (lane == 0 && ha.dbg) ha.dbg[dbg_blk * 4 + 0] = wall_clock64();
#endif
    const int cells = g.gh * g.gw;
    const int cell0 = (tile - g.tile_begin) * kFrontCells;
    const int cell = cell0 + 2 * lane;            // this lane's cells: cell, cell + 1
    const bool in0 = cell < cells, in1 = cell + 1 < cells;

    float t[2][NA][V];
    // The objectness planes are requested first: the filter and the ranking only need them,
    // so they run while the other 9 channels of every anchor are still in flight.  Whether
    // the 8-byte form can be used is decided once for the whole tile (cell0 is even, so with
    // an even plane size every plane slice starts on the parity of the head's base address);
    // otherwise lanes past the last cell read cell 0 of the slice instead of branching.
    const float *img = gdata + (size_t)bl * NA * V * cells + cell0;   // scalar
    const bool wide = ((reinterpret_cast<uintptr_t>(gdata) & 7) == 0) && ((cells & 1) == 0) &&
                      (cell0 + kFrontCells <= cells);
    if (wide) {
        // (loaded as one 64-bit integer: as two floats the optimiser folds this branch and the
        // other one into a single set of 4-byte loads with selected per-lane 64-bit addresses,
        // which costs 120 registers)
        auto load_channel = [&](int a, int v) {
            const u64 q = reinterpret_cast<const u64 *>(img + (size_t)(a * V + v) * cells)[lane];
            t[0][a][v] = __uint_as_float((unsigned)q);
            t[1][a][v] = __uint_as_float((unsigned)(q >> 32));
        };
#pragma unroll
        for (int a = 0; a < NA; ++a) load_channel(a, 4);
#pragma unroll
        for (int a = 0; a < NA; ++a) {
#pragma unroll
            for (int v = 0; v < V; ++v)
                if (v != 4) load_channel(a, v);
        }
    } else {
        const int i0 = in0 ? 2 * lane : 0, i1 = in1 ? 2 * lane + 1 : 0;
        auto load_channel = [&](int a, int v) {
            const float *pl = img + (size_t)(a * V + v) * cells;
            t[0][a][v] = pl[i0];
            t[1][a][v] = pl[i1];
        };
#pragma unroll
        for (int a = 0; a < NA; ++a) load_channel(a, 4);
#pragma unroll
        for (int a = 0; a < NA; ++a) {
#pragma unroll
            for (int v = 0; v < V; ++v)
                if (v != 4) load_channel(a, v);
        }
    }
#ifdef YP_LOADONLY   // diagnostic build: how long the loads alone take (results are garbage)
    {
        float acc = 0.0f;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int v = 0; v < V; ++v) acc += t[c][a][v];
        if (lane == 0) cb.tile_count[(size_t)b * cb.tpi + tile] = acc == 12345.678f ? 1 : 0;
        return;
    }
#endif
#ifdef YP_STAMPS
    if (lane == 0 && ha.dbg) {
        asm volatile("" ::"v"(t[0][0][4]), "v"(t[1][NA - 1][4]));
        ha.dbg[dbg_blk * 4 + 1] = wall_clock64();
    }
#endif
    bool pass[2][NA];
    u64 bal[2][NA];
    int total = 0;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const bool in = c ? in1 : in0;
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            bool p = in && (t[c][a][4] >= t_hi);
            if (in && !p && !(t[c][a][4] <= t_lo)) {
                const float obj = yp_sigmoid(t[c][a][4]);
                p = (obj > conf) && (obj != 0.0f);
            }
            pass[c][a] = p;
            bal[c][a] = __ballot(p);
            total += __popcll(bal[c][a]);
        }
    }
    // rank in anchor order (row = cell * NA + anchor): everything of the lanes below, then this
    // lane's earlier (cell, anchor) pairs
    const u64 below = lanes_below(lane);
    int rank = 0;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int a = 0; a < NA; ++a) rank += __popcll(bal[c][a] & below);
    // the staging area holds kStageWin survivors; a tile with more (rare) takes several rounds
    for (int base = 0; base < total; base += kStageWin) {
        int rk = rank - base;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                if (pass[c][a] && rk >= 0 && rk < kStageWin) {
                    float rec[kStride];
#pragma unroll
                    for (int v = 0; v < 5; ++v) rec[v] = t[c][a][v];
                    rec[5] = __int_as_float((a << 24) | (cell + c));
#pragma unroll
                    for (int j = 0; j < CS; ++j) rec[6 + j] = t[c][a][5 + j];
#pragma unroll
                    for (int j = 6 + CS; j < kStride; ++j) rec[j] = 0.0f;
                    float4 *st = reinterpret_cast<float4 *>(s_stage + rk * kStride);
#pragma unroll
                    for (int q = 0; q < kStride / 4; ++q)
                        st[q] = make_float4(rec[4 * q], rec[4 * q + 1], rec[4 * q + 2], rec[4 * q + 3]);
                }
                rk += pass[c][a] ? 1 : 0;
            }
        }
        // single wave: its LDS writes above reach the LDS before its reads below (in-order
        // queue); the wavefront-scope fence only stops the compiler from moving them across
        // each other
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#ifdef YP_STAMPS
        if (lane == 0 && ha.dbg && base == 0) ha.dbg[dbg_blk * 4 + 2] = wall_clock64();
#endif
        const int end = total < base + kStageWin ? total : base + kStageWin;
        for (int r = base + lane; r < end; r += 64) {
            float st[kStride];
            {
                const float4 *src = reinterpret_cast<const float4 *>(s_stage + (r - base) * kStride);
#pragma unroll
                for (int q = 0; q < kStride / 4; ++q) {
                    const float4 v4 = src[q];
                    st[4 * q] = v4.x; st[4 * q + 1] = v4.y; st[4 * q + 2] = v4.z; st[4 * q + 3] = v4.w;
                }
            }
            const int packed = __float_as_int(st[5]);
            const int sa = packed >> 24, scell = packed & 0xffffff;
            const int row = scell / g.gw, col = scell - row * g.gw;
            // Class score = largest sigmoid, class id = its first index.  The sigmoid is monotone
            // and strictly separates arguments kYpGap apart (yp_math.h), so when no other logit
            // lies within kYpGap below the largest one, one sigmoid decides; otherwise (rare) all
            // CS are evaluated as written in the reference.
            float xm = st[6];
            int bi = 0;
#pragma unroll
            for (int j = 1; j < CS; ++j)
                if (st[6 + j] > xm) { xm = st[6 + j]; bi = j; }
            const float gap_lo = xm - kYpGap;
            bool decided = (xm >= kYpGapLo) && (xm <= kYpGapHi);
#pragma unroll
            for (int j = 0; j < CS; ++j) decided = decided && !((st[6 + j] > gap_lo) && (st[6 + j] != xm));
            // the box, objectness and top class exponentials in one staged evaluation
            // (independent chains interleaved); sigmoid(x) = 1 / (1 + exp(-x)) as in yp_sigmoid
            float arg[6], e[6];
            arg[0] = -st[0];
            arg[1] = -st[1];
            arg[2] = st[2];
            arg[3] = st[3];
            arg[4] = -st[4];
            arg[5] = -xm;
            yp_exp_staged<6>(arg, e);
            const float obj = 1.0f / (1.0f + e[4]);
            const float cx = (1.0f / (1.0f + e[0]) + (float)col) * g.sx;
            const float cy = (1.0f / (1.0f + e[1]) + (float)row) * g.sy;
            const float w = (e[2] * g.anc[2 * sa + 0]) * g.sx;
            const float hgt = (e[3] * g.anc[2 * sa + 1]) * g.sy;
            float best = 1.0f / (1.0f + e[5]);
            if (!decided) {   // (a rolled loop over the staged record: this path must stay small)
                const float *lg = s_stage + (r - base) * kStride + 6;
                best = yp_sigmoid(lg[0]);
                bi = 0;
#pragma unroll 1
                for (int j = 1; j < CS; ++j) {
                    const float sc = yp_sigmoid(lg[j]);
                    if (sc > best) { best = sc; bi = j; }
                }
            }
            const float hw = w * 0.5f, hh = hgt * 0.5f;
            YpBox bx = {cx - hw, cy - hh, cx + hw, cy + hh};
            store_cand(cb, ((size_t)b * cb.tpi + tile) * cb.ts + r, bx, obj + 0.0f, best, bi);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();   // the next round overwrites the staging area
    }
    if (lane == 0) cb.tile_count[(size_t)b * cb.tpi + tile] = total;
    clear_ctrl(cb);
#ifdef YP_STAMPS
    if (lane == 0 && ha.dbg) ha.dbg[dbg_blk * 4 + 3] = wall_clock64();
#endif
}

// ---------------------------------------------------------------- front: reference head
// Thread t <-> row n = tile*256 + t of image b; a row is the V contiguous floats of one
// (row, col, anchor) (detect_net.py:153).
// CS > 0: the class count is known at compile time (the softmax exponentials are then held in
// registers and evaluated once); CS == 0 takes it from the argument.
template <bool STAGED, int CS>
__global__ void k_front_grid_ref(const float *pred, int H, int W, int A, int C_arg,
                                 AnchorTable at, CandBuf cb, float conf, float im_h,
                                 float im_w)
{
    const int C = CS ? CS : C_arg;
    __shared__ u64 s_bal[kRowTile / 64];
    extern __shared__ float s_tile[];
    static_assert(kRowTile == kTileRows, "one record per thread");
    const int tile = blockIdx.x, b = blockIdx.y;
    const int N = H * W * A, V = 5 + C;
    const int n = tile * kRowTile + threadIdx.x;
    const float *tile_src = pred + ((size_t)b * N + (size_t)tile * kRowTile) * V;
    if (STAGED) yp_tile_load(tile_src, (long long)(N - tile * kRowTile) * V, V, s_tile);
    bool pass = false;
    YpBox bx = {0.f, 0.f, 0.f, 0.f};
    float best = 0.f;
    int bi = 0;
    float t03[4] = {0.f, 0.f, 0.f, 0.f};   // the box logits, decoded for survivors only when staged
    constexpr int kRec = 7;                 // floats a survivor parks in the (re-used) tile buffer
    const bool dense = STAGED && V >= kRec; // (records narrower than that leave no room for it)
    // box of anchor n of the image from its four logits (detect_net.py:95-102, :133, :155-177)
    auto decode_box = [&](int n_, const float *tb) {
        const int a = n_ % A, k = n_ / A;
        const float gx = (float)(k % H), gy = (float)(k / H);   // detect_net.py:155-162
        const float cx = (yp_sigmoid(tb[0]) + gx) / (float)H;   // :166-168
        const float cy = (yp_sigmoid(tb[1]) + gy) / (float)W;
        const float w = (yp_exp(tb[2]) * at.anc[2 * a + 0]) / (float)H;  // :177
        const float hgt = (yp_exp(tb[3]) * at.anc[2 * a + 1]) / (float)W;
        const float hw = w * 0.5f, hh = hgt * 0.5f;
        YpBox r;
        r.a = (cy - hh) * im_h;
        r.b = (cx - hw) * im_w;
        r.c = (cy + hh) * im_h;
        r.d = (cx + hw) * im_w;
        return r;
    };
    if (n < N) {
        const float *t = STAGED ? s_tile + threadIdx.x * V : tile_src + (size_t)threadIdx.x * V;
        const float obj = yp_sigmoid(t[4]);                     // :171
        float m = t[5];                                         // softmax, :180
        for (int j = 1; j < C; ++j) m = t[5 + j] > m ? t[5 + j] : m;
        float s = 0.f;
        if (CS) {
            float e[CS ? CS : 1];
#pragma unroll
            for (int j = 0; j < CS; ++j) e[j] = yp_exp(t[5 + j] - m);
#pragma unroll
            for (int j = 0; j < CS; ++j) s = s + e[j];
            best = obj * (e[0] / s);                            // :107-109
#pragma unroll
            for (int j = 1; j < CS; ++j) {
                const float sc = obj * (e[j] / s);
                if (sc > best) { best = sc; bi = j; }
            }
        } else {
            for (int j = 0; j < C; ++j) s = s + yp_exp(t[5 + j] - m);
            best = obj * (yp_exp(t[5] - m) / s);                // :107-109
            for (int j = 1; j < C; ++j) {
                const float sc = obj * (yp_exp(t[5 + j] - m) / s);
                if (sc > best) { best = sc; bi = j; }
            }
        }
        pass = best >= conf;                                    // :111
        if (dense) {
#pragma unroll
            for (int q = 0; q < 4; ++q) t03[q] = t[q];
        } else {
            bx = decode_box(n, t);
        }
    }
    int total;
    const int rank = tile_rank_rows(pass, s_bal, total);   // (barrier inside: the tile has been read)
    const size_t slot0 = ((size_t)b * cb.tpi + tile) * cb.ts;
    if (dense) {
        // survivors park what their box needs; thread r then decodes the tile's r-th survivor, so
        // the four transcendentals of a box are evaluated for survivors only
        if (pass) {
            float *rec = s_tile + rank * kRec;
#pragma unroll
            for (int q = 0; q < 4; ++q) rec[q] = t03[q];
            rec[4] = __int_as_float(n);
            rec[5] = best;
            rec[6] = __int_as_float(bi);
        }
        __syncthreads();
        const int r = threadIdx.x;
        if (r < total) {
            const float *rec = s_tile + r * kRec;
            const YpBox rb = decode_box(__float_as_int(rec[4]), rec);
            store_cand(cb, slot0 + r, rb, rec[5], 0.f, __float_as_int(rec[6]));
        }
    } else if (pass) {
        store_cand(cb, slot0 + rank, bx, best, 0.f, bi);
    }
    if (threadIdx.x == 0) cb.tile_count[(size_t)b * cb.tpi + tile] = total;
    clear_ctrl(cb);
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
    clear_ctrl(cb);
}

// ------------------------------------------------------------ front: decoded YOLOv3 rows
template <bool STAGED>
__global__ void k_front_rows_v3(const float *rows, int N, int C, CandBuf cb, float conf)
{
    __shared__ u64 s_bal[kRowTile / 64];
    extern __shared__ float s_tile[];
    const int tile = blockIdx.x, b = blockIdx.y;
    const int V = 5 + C;
    const int n = tile * kRowTile + threadIdx.x;
    const float *tile_src = rows + ((size_t)b * N + (size_t)tile * kRowTile) * V;
    if (STAGED) yp_tile_load(tile_src, (long long)(N - tile * kRowTile) * V, V, s_tile);
    bool pass = false;
    YpBox bx = {0.f, 0.f, 0.f, 0.f};
    float best = 0.f, ob = 0.f;
    int bi = 0;
    if (n < N) {
        const float *p = STAGED ? s_tile + threadIdx.x * V : tile_src + (size_t)threadIdx.x * V;
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
    clear_ctrl(cb);
}

// --------------------------------------------------------------------------------- sorting
// Ascending bitonic sort of n_pad (power of two) 64-bit keys by the whole workgroup.
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

// Bitonic sort of exactly kSegThreads * E keys with E keys per thread held in registers
// (thread t owns positions t*E .. t*E+E-1).  Compare-exchange partners at distance j are in the
// same thread for j < E, in another lane of the wave for j < 64*E (xor-shuffles, no barrier) and
// only beyond that in another wave, where the pass goes through `keys` (LDS or global memory)
// with two barriers.  For 2048 keys that is 6 passes through memory instead of 66.
template <int E, class KeyPtr>
__device__ __forceinline__ void bitonic_sort_wide(KeyPtr keys, int n_pad)
{
    const int t = threadIdx.x, lane = t & 63;
    u64 v[E];
#pragma unroll
    for (int r = 0; r < E; ++r) v[r] = keys[t * E + r];
    for (int k = 2; k <= n_pad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64 * E) {
#pragma unroll
                for (int r = 0; r < E; ++r) keys[t * E + r] = v[r];
                __syncthreads();
#pragma unroll
                for (int r = 0; r < E; ++r) {
                    const int idx = t * E + r;
                    const u64 y = keys[idx ^ j];
                    const bool keep_min = ((idx & j) == 0) == ((idx & k) == 0);
                    v[r] = (keep_min == (v[r] < y)) ? v[r] : y;
                }
                __syncthreads();
            } else if (j >= E) {
                const int dl = j / E;   // partner lane = lane ^ dl, same register index
#pragma unroll
                for (int r = 0; r < E; ++r) {
                    const int idx = t * E + r;
                    const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v[r], dl, 64);
                    const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(v[r] >> 32), dl, 64);
                    const u64 y = ((u64)hi << 32) | lo;
                    const bool keep_min = ((lane & dl) == 0) == ((idx & k) == 0);
                    v[r] = (keep_min == (v[r] < y)) ? v[r] : y;
                }
            } else {
#pragma unroll
                for (int r = 0; r < E; ++r) {
                    if ((r & j) == 0) {   // (j < E: both of the pair are this thread's)
                        const int idx = t * E + r;
                        const u64 a = v[r], b = v[r | j];
                        const bool up = (idx & k) == 0;
                        const bool swap = (a > b) == up;
                        v[r] = swap ? b : a;
                        v[r | j] = swap ? a : b;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < E; ++r) keys[t * E + r] = v[r];
    __syncthreads();
}

// Stages k_first..k_last of the ascending bitonic network on ONE block of kSegThreads * E keys
// that sits in LDS and holds positions base .. base + kSegThreads*E - 1 of a longer sequence
// (base a multiple of the block size): the passes whose partner distance is below the block size,
// with the keys in registers as in bitonic_sort_wide.  The direction of a compare-exchange is
// that of the whole sequence's network, (position & k) == 0 -> ascending.
template <int E, class KeyPtr>
__device__ __forceinline__ void bitonic_block_wide(KeyPtr keys, int base, int k_first, int k_last)
{
    constexpr int NL = kSegThreads * E;
    const int t = threadIdx.x, lane = t & 63;
    u64 v[E];
#pragma unroll
    for (int r = 0; r < E; ++r) v[r] = keys[t * E + r];
    for (int k = k_first; k <= k_last; k <<= 1) {
        for (int j = (k >> 1) < (NL >> 1) ? (k >> 1) : (NL >> 1); j > 0; j >>= 1) {
            if (j >= 64 * E) {
#pragma unroll
                for (int r = 0; r < E; ++r) keys[t * E + r] = v[r];
                __syncthreads();
#pragma unroll
                for (int r = 0; r < E; ++r) {
                    const int idx = t * E + r;
                    const u64 y = keys[idx ^ j];
                    const bool keep_min = ((idx & j) == 0) == (((base + idx) & k) == 0);
                    v[r] = (keep_min == (v[r] < y)) ? v[r] : y;
                }
                __syncthreads();
            } else if (j >= E) {
                const int dl = j / E;   // partner lane = lane ^ dl, same register index
#pragma unroll
                for (int r = 0; r < E; ++r) {
                    const int idx = t * E + r;
                    const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v[r], dl, 64);
                    const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(v[r] >> 32), dl, 64);
                    const u64 y = ((u64)hi << 32) | lo;
                    const bool keep_min = ((lane & dl) == 0) == (((base + idx) & k) == 0);
                    v[r] = (keep_min == (v[r] < y)) ? v[r] : y;
                }
            } else {
#pragma unroll
                for (int r = 0; r < E; ++r) {
                    if ((r & j) == 0) {   // (j < E: both of the pair are this thread's)
                        const int idx = t * E + r;
                        const u64 a = v[r], b = v[r | j];
                        const bool up = ((base + idx) & k) == 0;
                        const bool swap = (a > b) == up;
                        v[r] = swap ? b : a;
                        v[r | j] = swap ? a : b;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < E; ++r) keys[t * E + r] = v[r];
    __syncthreads();
}

// Ascending bitonic sort of n_pad > kBigSortCap keys (a power of two) that live in global memory,
// by one workgroup: only the passes whose partner distance reaches the LDS block (kBigSortCap
// keys) go through global memory -- log2(n_pad / block) * (that + 1) / 2 of them, 6 for 32 768
// keys instead of 120 --, everything below is done block by block inside LDS.
__device__ __forceinline__ void bitonic_sort_blocked(u64 *g, u64 *lds, int n_pad)
{
    constexpr int BLK = kBigSortCap, E = kBigSortCap / kSegThreads;
    static_assert(E * kSegThreads == kBigSortCap && (E & (E - 1)) == 0, "a block is E keys per thread");
    const int tid = threadIdx.x;
    auto block_stages = [&](int base, int k_first, int k_last) {
        for (int i = tid; i < BLK; i += kSegThreads) lds[i] = g[base + i];
        __syncthreads();
        bitonic_block_wide<E>(lds, base, k_first, k_last);
        for (int i = tid; i < BLK; i += kSegThreads) g[base + i] = lds[i];
        __syncthreads();
    };
    for (int base = 0; base < n_pad; base += BLK) block_stages(base, 2, BLK);   // sorted blocks, alternating direction
    const int half = n_pad >> 1;
    for (int k = 2 * BLK; k <= n_pad; k <<= 1) {
        for (int j = k >> 1; j >= BLK; j >>= 1) {
            for (int t = tid; t < half; t += kSegThreads) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const u64 x = g[i], y = g[l];
                const bool up = (i & k) == 0;
                if ((x > y) == up) {
                    g[i] = y;
                    g[l] = x;
                }
            }
            __syncthreads();
        }
        for (int base = 0; base < n_pad; base += BLK) block_stages(base, k, k);
    }
}

// Rank sort of K <= kSmallCap distinct keys: element e counts the keys below it and lands
// at that position.  `in` is padded with ~0 up to a multiple of 8 keys.  When K is small,
// S = 2^k threads share one element (each counts a slice of the keys, combined with
// xor-shuffles) so the whole workgroup works; every lane of a wave reads the same LDS
// address pair at a time (broadcast reads).  One barrier at the end.
__device__ __forceinline__ void rank_sort(const u64 *in, u64 *out, int K)
{
    const int kp = (K + 7) & ~7;
    int S = 1;
    while (S < 8 && K * (S << 1) <= kSegThreads) S <<= 1;
    const int e = threadIdx.x / S, sub = threadIdx.x & (S - 1);
    const int len = ((kp / 8 + S - 1) / S) * 8;      // keys per slice, multiple of 8
    const int j0 = sub * len;
    const int j1 = (j0 + len) < kp ? (j0 + len) : kp;
    const u64 mine = e < K ? in[e] : 0ull;
    const ulonglong2 *in2 = reinterpret_cast<const ulonglong2 *>(in);
    int rank = 0;
    for (int j = j0; j < j1; j += 8) {
        const ulonglong2 a = in2[j >> 1], b = in2[(j >> 1) + 1], c = in2[(j >> 1) + 2],
                         d = in2[(j >> 1) + 3];
        rank += (a.x < mine) + (a.y < mine) + (b.x < mine) + (b.y < mine);
        rank += (c.x < mine) + (c.y < mine) + (d.x < mine) + (d.y < mine);
    }
    for (int off = 1; off < S; off <<= 1) rank += __shfl_xor(rank, off, 64);
    if (e < K && sub == 0) out[rank] = mine;
    __syncthreads();
}

__device__ __forceinline__ u64 uniform64(u64 v)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((u64)hi << 32) | lo;
}

__device__ __forceinline__ u64 readlane64(u64 v, int l)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((u64)hi << 32) | lo;
}

// ------------------------------------------------------------------------------ segment NMS
// Kept rows of up to kPend chunks wait in LDS before one pass over the later columns applies
// them all ("flush"); until then every new chunk is first tested against the waiting rows.
constexpr int kPend = 4;
constexpr int kPendLongList = 8192;   // segments above this use all kPend chunks of room

// LDS scratch of the chunk loop.
struct ChunkShared {
    float4 cbox[64];            // the chunk's 64 sorted rows as stored
    YpPrep cprep[64];           // the same rows prepared for IoU tests
    YpPrep kprep[kPend * 64];   // kept rows not yet applied to the later columns, compacted
    u64 sup[64];                // column c: alive earlier rows of the chunk that suppress row c
    u64 mask[2];                // [0] alive rows of the chunk, [1] kept rows of the chunk
    unsigned long long pend_sup;  // rows of the chunk suppressed by a pending kept row
};

// Greedy NMS over K sorted rows in chunks of 64.  sbox / removed / sslot are indexed by
// sorted position and live either in LDS (small segments) or in the global workspace.
// Returns the number of kept rows; emits each kept row through `emit(m, sorted_pos, box)`.
constexpr int kColUnroll = 4;

// Staged suppression test of one row per lane (pi, valid where row_ok) against U columns that
// are uniform over the wave: words[u] has bit l set when lane l's row suppresses column u.
// The arithmetic is yp_suppress_pretest's, written stage by stage over the U independent
// columns so that neighbouring instructions do not depend on each other; lanes the guard
// bands leave undecided take the exact divide-and-compare.
template <int MODE, int U>
__device__ __forceinline__ void suppress_words(const YpPrep &pi, bool row_ok, const float4 *cj,
                                               const float *aj, const YpThr &thr, u64 *words,
                                               unsigned int &ties)
{
    const float plus1 = MODE == YP_MODE_YOLOV3 ? 1.0f : 0.0f;
    float lo0[U], lo1[U], hi0[U], hi1[U], d0[U], d1[U], inter[U], uni[U];
#pragma unroll
    for (int u = 0; u < U; ++u) lo0[u] = vmax(pi.lo0, cj[u].x);
#pragma unroll
    for (int u = 0; u < U; ++u) lo1[u] = vmax(pi.lo1, cj[u].y);
#pragma unroll
    for (int u = 0; u < U; ++u) hi0[u] = vmin(pi.hi0, cj[u].z);
#pragma unroll
    for (int u = 0; u < U; ++u) hi1[u] = vmin(pi.hi1, cj[u].w);
#pragma unroll
    for (int u = 0; u < U; ++u) d0[u] = hi0[u] - lo0[u] + plus1;
#pragma unroll
    for (int u = 0; u < U; ++u) d1[u] = hi1[u] - lo1[u] + plus1;
#pragma unroll
    for (int u = 0; u < U; ++u) d0[u] = vmax(d0[u], 0.0f);
#pragma unroll
    for (int u = 0; u < U; ++u) d1[u] = vmax(d1[u], 0.0f);
#pragma unroll
    for (int u = 0; u < U; ++u) inter[u] = d0[u] * d1[u];
#pragma unroll
    for (int u = 0; u < U; ++u) uni[u] = pi.area + aj[u] - inter[u];
    // verdict per lane as a small integer (0 no, 1 yes, 2 undecided) kept in vector registers:
    // the select / or instructions cost vector issue slots but spare the scalar unit, which the
    // four SIMDs of a CU share and which the mask form of the same logic saturates first
    int verdict[U];
    int undecided = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const bool yes = inter[u] > thr.hi * uni[u];
        const bool no = inter[u] < thr.lo * uni[u];
        // a union above 1e30 needs no guard: the bands then decide "no", which is what the
        // quotient gives
        int v = (thr.fast & (uni[u] > 1e-30f) & (yes | no)) ? (yes ? 1 : 0) : 2;
        if (MODE == YP_MODE_REFERENCE)
            v = ((pi.area <= 0.0f) | (aj[u] <= 0.0f)) ? ((0.0f > thr.thr) ? 1 : 0) : v;
        v = row_ok ? v : 0;
        verdict[u] = v;
        undecided |= v;
    }
    if (__any(undecided & 2)) {   // rare: some quotient sits within 2^-20 of the threshold
        int nt = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            nt += __popcll(__ballot(verdict[u] == 2));
            if (verdict[u] == 2) {
                const YpPrep pj = {cj[u].x, cj[u].y, cj[u].z, cj[u].w, aj[u]};
                verdict[u] = yp_suppresses_exact<MODE>(pi, pj, thr.thr) ? 1 : 0;
            }
        }
        ties += (threadIdx.x & 63) == 0 ? (unsigned int)nt : 0u;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) words[u] = __ballot(verdict[u] == 1);
}

// two binary32 values that are added and multiplied together (one packed instruction on gfx950)
#if defined(__clang__)
typedef float yp_f2 __attribute__((ext_vector_type(2)));
#else
typedef float yp_f2 __attribute__((vector_size(8)));
#endif

// Two columns a, b against one row per lane, two tests per packed instruction.  was / wbs: bit l
// set when lane l's row suppresses the column (a, b may be uniform over the wave or differ from
// lane to lane).  The arithmetic is yp_suppress_pretest's for boxes whose
// union is known to be positive (x2 >= x1, y2 >= y1, finite areas: checked when the boxes were
// loaded) and a threshold inside the guard-band range; rows in ca_rows / cb_rows whose quotient
// the bands leave undecided take the exact divide-and-compare and are counted as ties.
template <int MODE>
__device__ __forceinline__ void pair2_words(const float4 &ri, float ai, const float4 &ca,
                                          const float4 &cb, float aa, float ab, const YpThr &thr,
                                          u64 ca_rows, u64 cb_rows, u64 &was, u64 &wbs,
                                          unsigned int &ties)
{
    const float plus1 = MODE == YP_MODE_YOLOV3 ? 1.0f : 0.0f;
    const yp_f2 lo0 = {vmax(ri.x, ca.x), vmax(ri.x, cb.x)};
    const yp_f2 lo1 = {vmax(ri.y, ca.y), vmax(ri.y, cb.y)};
    const yp_f2 hi0 = {vmin(ri.z, ca.z), vmin(ri.z, cb.z)};
    const yp_f2 hi1 = {vmin(ri.w, ca.w), vmin(ri.w, cb.w)};
    yp_f2 d0 = hi0 - lo0;
    yp_f2 d1 = hi1 - lo1;
    d0 = d0 + plus1;
    d1 = d1 + plus1;
    d0[0] = vmax(d0[0], 0.0f);
    d0[1] = vmax(d0[1], 0.0f);
    d1[0] = vmax(d1[0], 0.0f);
    d1[1] = vmax(d1[1], 0.0f);
    const yp_f2 inter = d0 * d1;
    const yp_f2 aj = {aa, ab};
    yp_f2 uni = ai + aj;
    uni = uni - inter;
    const yp_f2 t_hi = uni * thr.hi;
    const yp_f2 t_lo = uni * thr.lo;
    u64 ya = __ballot(inter[0] > t_hi[0]);
    u64 yb = __ballot(inter[1] > t_hi[1]);
    const u64 na = __ballot(inter[0] < t_lo[0]);
    const u64 nb = __ballot(inter[1] < t_lo[1]);
    const u64 ua = ~(ya | na) & ca_rows;
    const u64 ub = ~(yb | nb) & cb_rows;
    if ((ua | ub) != 0ull) {   // rare: some quotient sits within 2^-20 of the threshold
        const int lane = threadIdx.x & 63;
        const YpPrep pi = {ri.x, ri.y, ri.z, ri.w, ai};
        const YpPrep pa = {ca.x, ca.y, ca.z, ca.w, aa};
        const YpPrep pb = {cb.x, cb.y, cb.z, cb.w, ab};
        const bool ea = ((ua >> lane) & 1ull) && yp_suppresses_exact<MODE>(pi, pa, thr.thr);
        const bool eb = ((ub >> lane) & 1ull) && yp_suppresses_exact<MODE>(pi, pb, thr.thr);
        const u64 xa = __ballot(ea);
        const u64 xb = __ballot(eb);
        ya = (ya & ~ua) | xa;
        yb = (yb & ~ub) | xb;
        ties += (unsigned int)(__popcll(ua) + __popcll(ub));
    }
    was = ya & ca_rows;
    wbs = yb & cb_rows;
}

template <int MODE, class Emit>
__device__ __forceinline__ int nms_chunks(const float4 *sbox, unsigned char *removed, int K,
                                          const YpThr thr, int max_out, ChunkShared &sh,
                                          float4 *xraw, int *xidx, bool prune,
                                          unsigned int &pairs, unsigned int &ties,
                                          unsigned long long *dbgseg, Emit emit)
{
#ifdef YP_STAMPS
    unsigned long long acc[4] = {0, 0, 0, 0}, tprev = clock64();
#define CHUNK_T(k)                                   \
    do {                                             \
        const unsigned long long tn = clock64();     \
        acc[k] += tn - tprev;                        \
        tprev = tn;                                  \
    } while (0)
#else
#define CHUNK_T(k) do { } while (0)
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int nkept = 0;
    int npend = 0;   // kept rows waiting in sh.kprep (uniform over the workgroup)
    // a flush walks every live column, so very long lists let more rows wait before one
    const int pend_limit = K > kPendLongList ? (kPend - 1) * 64 : (kPend > 2 ? 64 : (kPend - 1) * 64);
    // the x-ordered list is split into one contiguous part per wave (whole batches of 64)
    const int per_wave = (((K + 63) >> 6) + kSegWaves - 1) / kSegWaves * 64;
    const int wbeg = wave * per_wave < K ? wave * per_wave : K;
    int wlen = (K - wbeg) < per_wave ? (K - wbeg) : per_wave;
    for (int r0 = 0; r0 < K && nkept < max_out; r0 += 64) {
        if (wave == 0) {
            const int i = r0 + lane;
            float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
            bool alive = false;
            if (i < K) {
                bx = sbox[i];
                alive = removed[i] == 0;
            }
            const YpBox bl = {bx.x, bx.y, bx.z, bx.w};
            sh.cbox[lane] = bx;
            sh.cprep[lane] = yp_prep<MODE>(bl);
            const u64 alive0 = __ballot(alive);
            if (lane == 0) {
                sh.mask[0] = alive0;
                sh.pend_sup = 0ull;
            }
        }
        __syncthreads();
        CHUNK_T(0);
        u64 alive_mask = uniform64(sh.mask[0]);
        // the waiting kept rows against the chunk: lanes hold the chunk's rows, every wave
        // walks its share of the waiting rows four at a time, verdicts are OR-ed in LDS
        if (npend > 0 && alive_mask != 0ull) {
            const YpPrep pc = sh.cprep[lane];
            const bool col_ok = (alive_mask >> lane) & 1ull;
            u64 acc = 0ull;
            int tested = 0;
            for (int q0 = wave * 4; q0 < npend; q0 += kSegWaves * 4) {
                tested += npend - q0 < 4 ? npend - q0 : 4;
                float4 cj[4];
                float aj[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const YpPrep pr = sh.kprep[q0 + u < npend ? q0 + u : q0];
                    cj[u] = make_float4(pr.lo0, pr.lo1, pr.hi0, pr.hi1);
                    aj[u] = pr.area;
                }
                u64 words[4];
                suppress_words<MODE, 4>(pc, col_ok, cj, aj, thr, words, ties);
#pragma unroll
                for (int u = 0; u < 4; ++u) acc |= q0 + u < npend ? words[u] : 0ull;
            }
            if (lane == 0) {
                if (acc != 0ull) atomicOr(&sh.pend_sup, (unsigned long long)acc);
                pairs += (unsigned int)(__popcll(alive_mask) * tested);
            }
            __syncthreads();
            alive_mask &= ~uniform64(sh.pend_sup);
        }
        int nkc = 0;
        if (alive_mask != 0ull) {
            // 64x64 tile of the chunk against itself: lanes hold the rows, wave w tests the
            // columns 8w..8w+7 in staged passes; sup[c] = alive rows before c that suppress c
            {
                constexpr int kTileCols = 64 / kSegWaves;   // columns of the tile per wave
                static_assert(kTileCols % 4 == 0 && kTileCols * kSegWaves == 64, "whole passes of 4");
                const YpPrep pi = sh.cprep[lane];
                const bool row_ok = (alive_mask >> lane) & 1ull;
                u64 mine = 0ull;
#pragma unroll
                for (int part = 0; part < kTileCols / 4; ++part) {
                    float4 cj[4];
                    float aj[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const YpPrep pc = sh.cprep[wave * kTileCols + 4 * part + u];
                        cj[u] = make_float4(pc.lo0, pc.lo1, pc.hi0, pc.hi1);
                        aj[u] = pc.area;
                    }
                    u64 words[4];
                    suppress_words<MODE, 4>(pi, row_ok, cj, aj, thr, words, ties);
#pragma unroll
                    for (int u = 0; u < 4; ++u) mine = lane == 4 * part + u ? words[u] : mine;
                }
                if (lane < kTileCols) {
                    const int c = wave * kTileCols + lane;
                    sh.sup[c] = mine & lanes_below(c);
                }
                if (tid == 0) {
                    const unsigned int na = (unsigned int)__popcll(alive_mask);
                    pairs += na * (na - 1u) / 2u;
                }
            }
            __syncthreads();
            CHUNK_T(1);
            // greedy keep set of the chunk as the fixed point of
            // keep[c] = alive[c] && no kept earlier row of the chunk suppresses c  (wave 0)
            if (wave == 0) {
                const u64 mycol = sh.sup[lane];
                const bool alive = (alive_mask >> lane) & 1ull;
                u64 keptm = alive_mask;
                for (int it = 0; it < 66; ++it) {
                    const u64 nk = __ballot(alive && (mycol & keptm) == 0ull);
                    if (nk == keptm) break;
                    keptm = nk;
                }
                const int room = max_out - nkept;
                if (__popcll(keptm) > room)
                    keptm = __ballot(((keptm >> lane) & 1ull) && __popcll(keptm & lanes_below(lane)) < room);
                if (lane == 0) sh.mask[1] = keptm;
                if ((keptm >> lane) & 1ull) {
                    const int pos = __popcll(keptm & lanes_below(lane));
                    sh.kprep[npend + pos] = sh.cprep[lane];
                    emit(nkept + pos, r0 + lane, sh.cbox[lane]);
                }
            }
            __syncthreads();
            CHUNK_T(2);
            nkc = __popcll(uniform64(sh.mask[1]));
            nkept += nkc;
            npend += nkc;
            if (nkept >= max_out) break;
        }
        // flush: the later columns are tested against the waiting kept rows once there is no
        // room for another chunk's worth of them.  Lanes hold columns, 64 neighbours of the
        // x-ordered list at a time, so a batch covers a narrow band of the image; only the
        // kept rows whose extent along the pruning axis reaches that band (2 px of slack:
        // more than any rounding of the IoU arithmetic) are walked, four at a time through
        // the staged test with the row values uniform over the wave.  A kept row outside the
        // band has intersection 0 with every column of the batch, which `prune` (threshold
        // inside the guard-band range, all areas positive) makes a "does not suppress"
        // verdict; without `prune` every waiting row is walked.  Every wave owns one contiguous
        // part of the list for the whole loop; a removed column is flagged in `removed` for
        // the chunk loader and leaves the list.
        if (npend > pend_limit && r0 + 64 < K) {
            const int nrw = (npend + 63) >> 6;   // words of waiting rows
            // this wave's part of the list is [wbeg, wbeg + wlen); survivors are written back
            // densely from wbeg on (never ahead of the read position), so the next flush finds
            // full batches and no struck or already processed columns
            int out = wbeg;
            const int wend = wbeg + wlen;
            int idx_c = -1;
            float4 raw_c = make_float4(0.f, 0.f, 0.f, 0.f);
            if (wbeg + lane < wend) {
                idx_c = xidx[wbeg + lane];
                raw_c = xraw[wbeg + lane];
            }
            for (int pos = wbeg; pos < wend; pos += 64) {
                const int pn = pos + 64 + lane;   // this lane's position in the next batch
                int idx_n = -1;
                float4 raw_n = make_float4(0.f, 0.f, 0.f, 0.f);
                if (pn < wend) {
                    idx_n = xidx[pn];
                    raw_n = xraw[pn];
                }
                const bool live = idx_c >= r0 + 64;
                const u64 livem = __ballot(live);
                if (livem != 0ull) {
                    const YpBox bj = {raw_c.x, raw_c.y, raw_c.z, raw_c.w};
                    const YpPrep pj = yp_prep<MODE>(bj);
                    const float clo = MODE == YP_MODE_YOLOV3 ? pj.lo0 : pj.lo1;
                    const float chi = MODE == YP_MODE_YOLOV3 ? pj.hi0 : pj.hi1;
                    float bl = live ? clo : __builtin_inff();
                    float bh = live ? chi : -__builtin_inff();
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) {
                        bl = vmin(bl, __shfl_xor(bl, off, 64));
                        bh = vmax(bh, __shfl_xor(bh, off, 64));
                    }
                    u64 drop = 0ull;
                    for (int rw = 0; rw < nrw; ++rw) {
                        const int rq = (rw << 6) + lane;
                        const YpPrep kr = sh.kprep[rq < npend ? rq : 0];
                        const float rlo = MODE == YP_MODE_YOLOV3 ? kr.lo0 : kr.lo1;
                        const float rhi = MODE == YP_MODE_YOLOV3 ? kr.hi0 : kr.hi1;
                        // (negated comparisons: a NaN keeps the row)
                        u64 todo = __ballot(rq < npend &&
                                            (!prune || (!(rhi + 2.0f < bl) && !(rlo - 2.0f > bh))));
                        if (lane == 0) pairs += (unsigned int)(__popcll(livem) * __popcll(todo));
                        while (todo != 0ull) {
                            int q[kColUnroll];
                            float4 cj[kColUnroll];
                            float aj[kColUnroll];
#pragma unroll
                            for (int u = 0; u < kColUnroll; ++u) {
                                q[u] = todo != 0ull ? __ffsll((long long)todo) - 1 : -1;
                                todo &= todo - 1ull;   // 0 stays 0
                                const YpPrep pr = sh.kprep[(rw << 6) + (q[u] < 0 ? 0 : q[u])];
                                cj[u] = make_float4(pr.lo0, pr.lo1, pr.hi0, pr.hi1);
                                aj[u] = pr.area;
                            }
                            u64 words[kColUnroll];
                            suppress_words<MODE, kColUnroll>(pj, live, cj, aj, thr, words, ties);
#pragma unroll
                            for (int u = 0; u < kColUnroll; ++u) drop |= q[u] >= 0 ? words[u] : 0ull;
                        }
                    }
                    const bool dropped = (drop >> lane) & 1ull;   // (only live lanes can be set)
                    if (dropped) removed[idx_c] = 1;
                    const u64 stay = livem & ~drop;
                    if (live && !dropped) {
                        const int dst = out + __popcll(stay & lanes_below(lane));
                        xidx[dst] = idx_c;
                        xraw[dst] = raw_c;
                    }
                    out += __popcll(stay);
                }
                idx_c = idx_n;
                raw_c = raw_n;
            }
            wlen = out - wbeg;
            npend = 0;
        }
        __syncthreads();
        CHUNK_T(3);
    }
#ifdef YP_STAMPS
    if (threadIdx.x == 0 && dbgseg)
        for (int k = 0; k < 4; ++k) dbgseg[24 + k] = acc[k];
#endif
    return nkept;
}

#include "yp_graph.h"
#include "yp_image.h"

template <int MODE>
__device__ __forceinline__ void segment_nms_body(const SegArgs &s, const int vblk, const int nblk)
{
    [[maybe_unused]] const int stamp_block = vblk;   // (diagnostic builds)
    // s_raw: sort keys of segments above kSmallCap (bitonic sort; up to kBigSortCap keys run on into s_small); for
    // small segments the triangular pairwise suppression bitmask.
    // s_small: [0, 8(cap+8)) the sorted slot ids and box areas;
    // [8(cap+8), 16(cap+8)) sorted keys; [16(cap+8), 32(cap+8)) the sorted boxes.
    // (one block, so that a segment of up to kBigSortCap keys can be sorted across both parts)
    __shared__ __align__(16) unsigned char s_lds[kRawBytes + kSmallBytes];
    static_assert(kRawBytes % 16 == 0 && kRawBytes + kSmallBytes >= kBigSortCap * 8, "layout");
    unsigned char *s_raw = s_lds, *s_small = s_lds + kRawBytes;
    __shared__ ChunkShared s_chunk;
    __shared__ int s_cnt[4];   // [0] candidates of lower classes, [1] of this class
    __shared__ int s_scan[kSegWaves];
    __shared__ unsigned int s_pairs, s_ties;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Workgroups b and b + 8 share an XCD (and its L2): deal every XCD a contiguous run of
    // segments, so that the classes of one image -- which all read that image's candidate
    // arrays -- are fetched into one L2 instead of up to C of them.  (Speed only: any
    // bijection of workgroups to segments is correct.)
    int seg;
    {
        const int x = vblk & 7, k = vblk >> 3;
        const int per = nblk >> 3, rem = nblk & 7;
        seg = x < rem ? x * (per + 1) + k : rem * (per + 1) + (x - rem) * per + k;
    }
    const int b = s.per_class ? seg / s.C : seg;
    const int c = s.per_class ? seg - b * s.C : 0;
    const size_t ibase = (size_t)b * s.spi;
    const int *tcount = s.tile_count + (size_t)b * s.tpi;
    int *gpre = s.tile_prefix + (size_t)b * s.tpi;
    const int tpi = s.tpi, ts = s.ts;

    if (tid < 4) s_cnt[tid] = 0;
    if (tid == 0) s_pairs = 0u;
    if (tid == 1) s_ties = 0u;
    if (MODE == YP_MODE_YOLOV3 && tid == 0 && s.gseg) s.gseg[seg].K = 0;
    // an image the per-image kernel has finished needs nothing here
    if (MODE == YP_MODE_YOLOV3 && s.img_flag != nullptr && s.img_flag[b] == 0) return;
    STAMP(0);

    // ---- exclusive prefix of the image's tile counts: a kept row's position in the image's
    // survivor list, which only the reference-mode output reports
    if (MODE == YP_MODE_REFERENCE) {
        int k_img = 0;
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
            if (base + tid < tpi) gpre[base + tid] = k_img + woff + incl - v;
            k_img += total;
            __syncthreads();
        }
    } else {
        __syncthreads();   // s_cnt is cleared
    }

    STAMP(1);
    // ---- gather: a wave takes kGatherTiles tiles per round, lane r looks at candidate r of each
    // (the tile's count and the candidate's key and class are requested together, so a round is
    // one trip to memory; slots past the count hold stale values that are ignored), keeps the
    // candidates of this segment and appends their sort keys in any order
    u64 *keys_in = reinterpret_cast<u64 *>(s_raw);
    // (the copy the small-segment path sorts from lives in the chunk loop's scratch, which nothing
    // else uses before that sort is done: the gather may then fill the WHOLE sort block, and a
    // segment of up to kBigSortCap candidates is gathered once)
    u64 *keys_small = reinterpret_cast<u64 *>(&s_chunk);
    static_assert(sizeof(ChunkShared) >= (kSmallCap + 8) * sizeof(u64), "the small path's keys fit");
    u64 *gk = s.gkeys + 2 * ibase;  // rebased to the segment's region below
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    auto gather = [&](bool to_lds, u64 *gdst) {
        int lt = 0;
        constexpr int kGatherTiles = 4;
        // kGatherTiles x 64 candidates whose key and class were requested together: those of this
        // segment get consecutive output positions from one reservation
        auto place = [&](const int (&slot)[kGatherTiles], const bool (&valid)[kGatherTiles],
                         const float (&kf)[kGatherTiles], const int (&cl)[kGatherTiles]) {
            u64 m[kGatherTiles];
            int total = 0;
#pragma unroll
            for (int u = 0; u < kGatherTiles; ++u) {
                lt += __popcll(__ballot(valid[u] && cl[u] < c));   // wave-uniform count
                m[u] = __ballot(valid[u] && cl[u] == c);
                total += __popcll(m[u]);
            }
            if (total > 0) {
                int basepos = 0;
                if (lane == 0) basepos = atomicAdd(&s_cnt[1], total);
                basepos = __shfl(basepos, 0, 64);
#pragma unroll
                for (int u = 0; u < kGatherTiles; ++u) {
                    if ((m[u] >> lane) & 1ull) {
                        const u64 key = ((u64)(~yp_float_order_bits(kf[u])) << 32) | (u64)(unsigned int)slot[u];
                        const int pos = basepos + __popcll(m[u] & lanes_below(lane));
                        if (to_lds) {
                            if (pos < kBigSortCap) keys_in[pos] = key;
                            if (pos < kSmallCap) keys_small[pos] = key;
                        } else {
                            gdst[pos] = key;
                        }
                    }
                    basepos += __popcll(m[u]);
                }
            }
        };
        // (tiles are dealt to the waves round-robin, so a few tiles still spread over the waves)
        for (int t0 = wv; t0 < tpi; t0 += kSegWaves * kGatherTiles) {
            int cnt[kGatherTiles];
            {
                int slot[kGatherTiles], cl[kGatherTiles];
                bool valid[kGatherTiles];
                float kf[kGatherTiles];
#pragma unroll
                for (int u = 0; u < kGatherTiles; ++u) {
                    const int t = t0 + u * kSegWaves;
                    const int tile = t < tpi ? t : tpi - 1;
                    slot[u] = tile * ts + lane;
                    cnt[u] = t < tpi ? tcount[tile] : 0;
                    kf[u] = s.cand_key[ibase + slot[u]];
                    cl[u] = s.per_class ? s.cand_cls[ibase + slot[u]] : c;
                }
#pragma unroll
                for (int u = 0; u < kGatherTiles; ++u) valid[u] = lane < cnt[u];
                place(slot, valid, kf, cl);
            }
            // tiles with more than 64 candidates (low confidence thresholds, crowded scenes): the
            // rest of each, again kGatherTiles x 64 candidates per trip to memory
#pragma nounroll
            for (int u = 0; u < kGatherTiles; ++u) {
                const int tbase = (t0 + u * kSegWaves) * ts;
                const int cu = u == 0 ? cnt[0] : (u == 1 ? cnt[1] : (u == 2 ? cnt[2] : cnt[3]));
                static_assert(kGatherTiles == 4, "the selection above");
                for (int r0 = 64; r0 < cu; r0 += 64 * kGatherTiles) {
                    int slot[kGatherTiles], cl[kGatherTiles];
                    bool valid[kGatherTiles];
                    float kf[kGatherTiles];
#pragma unroll
                    for (int v = 0; v < kGatherTiles; ++v) {
                        const int r = r0 + 64 * v + lane;
                        valid[v] = r < cu;
                        slot[v] = tbase + (valid[v] ? r : 0);
                        kf[v] = s.cand_key[ibase + slot[v]];
                        cl[v] = s.per_class ? s.cand_cls[ibase + slot[v]] : c;
                    }
                    place(slot, valid, kf, cl);
                }
            }
        }
        if (lane == 0 && lt) atomicAdd(&s_cnt[0], lt);
    };
    gather(true, nullptr);
    __syncthreads();
    STAMP(2);
    const int K = s_cnt[1];
    const int region = s_cnt[0];
    const size_t reg = ibase + region;
    if (tid == 0) {
        s.seg_cand[seg] = K;
        s.seg_region[seg] = region;
    }
    if (K == 0) {
        if (tid == 0) {
            s.seg_kept[seg] = 0;
            s.seg_pairs[seg] = 0u;
            s.seg_ties[seg] = 0u;
            if (MODE == YP_MODE_REFERENCE) {
                s.out_counts[b] = 0;
                s.surv_counts[b] = 0;
            }
        }
        return;
    }

    const int max_out = s.max_out > 0 ? s.max_out : 0x7fffffff;
    unsigned int pairs = 0, ties = 0;
    int nkept = 0;
    const int *sslot_final;  // sorted position -> slot id, readable after the NMS loop

    auto emit_kept = [&](const int *sslot, int m, int sorted_pos, const float4 &bx) {
        const int slot = sslot[sorted_pos];
        if (MODE == YP_MODE_REFERENCE) {
            const size_t q = (size_t)b * s.max_out + m;
            reinterpret_cast<float4 *>(s.out_boxes)[q] = bx;
            s.out_scores[q] = s.cand_key[ibase + slot];
            s.out_classes[q] = s.cand_cls[ibase + slot];
            const int tile = slot / ts;
            s.out_index[q] = gpre[tile] + (slot - tile * ts);
        } else {
            s.kept_slot[reg + m] = slot;
        }
    };

    if (K <= kSmallCap) {
        // ---- small segment: rank sort, then the full pairwise suppression bitmask in LDS
        // (triangular, word-major: word w of column j >= 64w holds the rows 64w..64w+63 that
        // precede j in sorted order and suppress it) and a sweep by wave 0 that needs no
        // barriers and no IoU arithmetic.
        u64 *cmask = reinterpret_cast<u64 *>(s_raw);
        u64 *keys_out = reinterpret_cast<u64 *>(s_small + 8 * (kSmallCap + 8));
        float4 *lbox = reinterpret_cast<float4 *>(s_small + 16 * (kSmallCap + 8));
        int *lslot = reinterpret_cast<int *>(s_small);
        float *larea = reinterpret_cast<float *>(s_small + 4 * kSmallCap);
        for (int i = K + tid; i < ((K + 7) & ~7); i += kSegThreads) keys_small[i] = ~0ull;
        __syncthreads();
        rank_sort(keys_small, keys_out, K);
        STAMP(3);
        for (int i = tid; i < K; i += kSegThreads) {
            const int slot = (int)(unsigned int)(keys_out[i] & 0xffffffffull);
            const float4 raw = s.cand_box[ibase + slot];
            const YpBox bx = {raw.x, raw.y, raw.z, raw.w};
            const YpPrep pp = yp_prep<MODE>(bx);   // reference mode: corners min/max-normalised
            lslot[i] = slot;
            lbox[i] = make_float4(pp.lo0, pp.lo1, pp.hi0, pp.hi1);
            larea[i] = pp.area;
        }
        const int nwords = (K + 63) >> 6;
        __syncthreads();
        STAMP(4);
        const YpThr thr = yp_make_thr(s.thr);
        const float plus1 = MODE == YP_MODE_YOLOV3 ? 1.0f : 0.0f;
        // all-pairs tile: lanes hold the 64 rows of word w; each wave walks the columns
        // j >= 64w it owns, kPairUnroll at a time, written stage by stage over the independent
        // columns so that neighbouring instructions do not depend on each other.  Same
        // arithmetic and guard bands as yp_suppress_pretest (one-instruction min/max);
        // undecided lanes take the exact divide-and-compare.
        constexpr int kPairUnroll = 3;
        const int wu = __builtin_amdgcn_readfirstlane(wave);
        // mask words of the columns [cbeg, cend) against the row words [0, wend)
        auto build_mask = [&](int cbeg, int cend, int wend) {
        for (int w = 0; w < wend; ++w) {
            const int i = (w << 6) + lane;
            const float4 ci = lbox[i < K ? i : 0];
            const float ai = larea[i < K ? i : 0];
            const YpPrep pi = {ci.x, ci.y, ci.z, ci.w, ai};
            u64 *cw = cmask + (w * K - 32 * w * (w - 1)) - (w << 6);   // cw[j], j >= 64w
            for (int j0 = ((w << 6) > cbeg ? (w << 6) : cbeg) + wu; j0 < cend;
                 j0 += kPairUnroll * kSegWaves) {
                int verdict[kPairUnroll];
                int undecided = 0;
                float4 cj[kPairUnroll];
                float aj[kPairUnroll], lo0[kPairUnroll], lo1[kPairUnroll], hi0[kPairUnroll],
                    hi1[kPairUnroll], d0[kPairUnroll], d1[kPairUnroll], inter[kPairUnroll],
                    uni[kPairUnroll];
                // columns j0, j0+8, ... at fixed offsets from one address; a column past K reads
                // whatever the padded arrays hold and its word is never stored
                const float4 *cb = lbox + j0;
                const float *ab = larea + j0;
#pragma unroll
                for (int u = 0; u < kPairUnroll; ++u) {
                    cj[u] = cb[u * kSegWaves];
                    aj[u] = ab[u * kSegWaves];
                }
#pragma unroll
                for (int u = 0; u < kPairUnroll; ++u) lo0[u] = vmax(ci.x, cj[u].x);
#pragma unroll
                for (int u = 0; u < kPairUnroll; ++u) lo1[u] = vmax(ci.y, cj[u].y);
#pragma unroll
                for (int u = 0; u < kPairUnroll; ++u) hi0[u] = vmin(ci.z, cj[u].z);
#pragma unroll
                for (int u = 0; u < kPairUnroll; ++u) hi1[u] = vmin(ci.w, cj[u].w);
#pragma unroll
                for (int u = 0; u < kPairUnroll; ++u) d0[u] = hi0[u] - lo0[u] + plus1;
#pragma unroll
                for (int u = 0; u < kPairUnroll; ++u) d1[u] = hi1[u] - lo1[u] + plus1;
#pragma unroll
                for (int u = 0; u < kPairUnroll; ++u) d0[u] = vmax(d0[u], 0.0f);
#pragma unroll
                for (int u = 0; u < kPairUnroll; ++u) d1[u] = vmax(d1[u], 0.0f);
#pragma unroll
                for (int u = 0; u < kPairUnroll; ++u) inter[u] = d0[u] * d1[u];
#pragma unroll
                for (int u = 0; u < kPairUnroll; ++u) uni[u] = ai + aj[u] - inter[u];
#pragma unroll
                for (int u = 0; u < kPairUnroll; ++u) {
                    const bool yes = inter[u] > thr.hi * uni[u];
                    const bool no = inter[u] < thr.lo * uni[u];
                    // a union above 1e30 needs no guard: the bands then decide "no", which
                    // is what the quotient gives
                    int v = (thr.fast & (uni[u] > 1e-30f) & (yes | no)) ? (yes ? 1 : 0) : 2;
                    if (MODE == YP_MODE_REFERENCE)
                        v = ((ai <= 0.0f) | (aj[u] <= 0.0f)) ? ((0.0f > thr.thr) ? 1 : 0) : v;
                    verdict[u] = v;
                    undecided |= v;
                }
                if (__any(undecided & 2)) {   // rare: some quotient sits within 2^-20 of the threshold
                    int nt = 0;
#pragma unroll
                    for (int u = 0; u < kPairUnroll; ++u) {
                        // (only the pairs of the triangle count: rows that precede the column)
                        const int ju = j0 + u * kSegWaves, before_u = ju - (w << 6);
                        const u64 rows_u = before_u >= 64 ? ~0ull : lanes_below(before_u);
                        nt += ju < cend ? __popcll(__ballot(verdict[u] == 2 && i < K) & rows_u) : 0;
                        if (verdict[u] == 2) {
                            const YpPrep pj = {cj[u].x, cj[u].y, cj[u].z, cj[u].w, aj[u]};
                            verdict[u] = yp_suppresses_exact<MODE>(pi, pj, thr.thr) ? 1 : 0;
                        }
                    }
                    ties += lane == 0 ? (unsigned int)nt : 0u;
                }
                // lane u stores the word of column u: one store instead of kPairUnroll
                // single-lane ones
                u64 mine = 0ull;
#pragma unroll
                for (int u = 0; u < kPairUnroll; ++u) {
                    const u64 word = __ballot(verdict[u] == 1);
                    mine = lane == u ? word : mine;
                }
                {
                    const int j = j0 + lane * kSegWaves;
                    const int before = j - (w << 6);       // rows of this word that precede column j
                    const u64 rows = before >= 64 ? ~0ull : lanes_below(before);
                    if (lane < kPairUnroll && j < cend) cw[j] = mine & rows;
                }
            }
        }
        };
        if (MODE == YP_MODE_REFERENCE && max_out < K) {
            // An output cap can end the sweep long before the last candidate: build the mask one
            // 64-column chunk at a time (against all earlier rows), sweep that chunk, and stop
            // once max_out rows are kept.  Kept words of earlier chunks wait in LDS.
            static_assert(sizeof(s_chunk.sup) >= kSmallWords * sizeof(u64), "kept words fit");
            u64 *s_keptw = s_chunk.sup;           // written and read by wave 0 only
            long long built = 0;
            for (int cwd = 0; cwd < nwords; ++cwd) {
                const int cbeg = cwd << 6, cend = cbeg + 64 < K ? cbeg + 64 : K;
                build_mask(cbeg, cend, cwd + 1);
                built += (long long)(cend - cbeg) * cbeg + (long long)(cend - cbeg) * (cend - cbeg - 1) / 2;
                __syncthreads();
                if (wave == 0) {
                    const int j = cbeg + lane;
                    const int jj = j < K ? j : cbeg;
                    bool alive = j < K;
                    for (int w = 0; w < cwd; ++w)
                        alive = alive && (cmask[(w * K - 32 * w * (w - 1)) + jj - (w << 6)] & s_keptw[w]) == 0ull;
                    const u64 mycol = (j < K) ? cmask[(cwd * K - 32 * cwd * (cwd - 1)) + jj - cbeg] : 0ull;
                    u64 keep = __ballot(alive);
                    for (int it = 0; it < 66; ++it) {
                        const u64 nk = __ballot(alive && (mycol & keep) == 0ull);
                        if (nk == keep) break;
                        keep = nk;
                    }
                    const int room = max_out - nkept;
                    if (__popcll(keep) > room)
                        keep = __ballot(((keep >> lane) & 1ull) && __popcll(keep & lanes_below(lane)) < room);
                    if ((keep >> lane) & 1ull)
                        emit_kept(lslot, nkept + __popcll(keep & lanes_below(lane)), j,
                                  s.cand_box[ibase + lslot[j]]);
                    nkept += __popcll(keep);
                    if (lane == 0) {
                        s_keptw[cwd] = keep;
                        s_cnt[3] = nkept;
                    }
                }
                __syncthreads();
                if (s_cnt[3] >= max_out) break;
            }
            pairs = tid == 0 ? (unsigned int)built : 0u;
        } else {
        build_mask(0, K, nwords);
        pairs = tid == 0 ? (unsigned int)((long long)K * (K - 1) / 2) : 0u;
        __syncthreads();
        STAMP(7);
        if (wave == 0) {
            u64 keptw[kSmallWords];
#pragma unroll
            for (int w = 0; w < kSmallWords; ++w) keptw[w] = 0ull;
#pragma unroll
            for (int cwd = 0; cwd < kSmallWords; ++cwd) {
                if (cwd < nwords && nkept < max_out) {
                    const int j = (cwd << 6) + lane;
                    const int jj = j < K ? j : (cwd << 6);
                    bool alive = j < K;
#pragma unroll
                    for (int w = 0; w < cwd; ++w)
                        alive = alive && (cmask[(w * K - 32 * w * (w - 1)) + jj - (w << 6)] & keptw[w]) == 0ull;
                    const u64 mycol = (j < K) ? cmask[(cwd * K - 32 * cwd * (cwd - 1)) + jj - (cwd << 6)] : 0ull;
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
                        emit_kept(lslot, nkept + __popcll(keep & lanes_below(lane)), j,
                                  s.cand_box[ibase + lslot[j]]);
                    nkept += __popcll(keep);
                }
            }
        }
        }
        sslot_final = lslot;
    } else {
        // ---- large segment: bitonic sort (LDS up to kBigSortCap keys, else global), sorted
        // boxes / flags / slots in the global workspace region of the segment
        int n_pad = 2;
        while (n_pad < K) n_pad <<= 1;
        const bool in_lds = K <= kBigSortCap;
        u64 *gdst = gk + 2 * (size_t)region;
        if (!in_lds) {
            __syncthreads();
            if (tid == 0) s_cnt[1] = 0;
            if (tid == 1) s_cnt[0] = 0;
            __syncthreads();
            gather(false, gdst);
        }
        for (int i = K + tid; i < n_pad; i += kSegThreads) {
            if (in_lds) keys_in[i] = ~0ull;
            else gdst[i] = ~0ull;
        }
        __syncthreads();
        if (in_lds && n_pad == 2 * kSegThreads) bitonic_sort_wide<2>(keys_in, n_pad);
        else if (in_lds && n_pad == 4 * kSegThreads) bitonic_sort_wide<4>(keys_in, n_pad);
        else if (in_lds && n_pad == 8 * kSegThreads) bitonic_sort_wide<8>(keys_in, n_pad);
        else if (in_lds) bitonic_sort(keys_in, n_pad);
        else bitonic_sort_blocked(gdst, keys_in, n_pad);
        STAMP(3);
        float4 *gbox = s.sbox + reg;
        int *gslot = s.sslot + reg;
        unsigned char *grem = s.removed + reg;
        for (int i = tid; i < K; i += kSegThreads) {
            const int slot = (int)(unsigned int)((in_lds ? keys_in[i] : gdst[i]) & 0xffffffffull);
            gslot[i] = slot;
            gbox[i] = s.cand_box[ibase + slot];
            grem[i] = 0;
        }
        __syncthreads();
        STAMP(4);
        sslot_final = gslot;
        // ---- graph path: a segment this large is not swept here.  It gets a spatial index (boxes
        // binned by size class and by the cell of their centre, see yp_graph.h) and is finished
        // by k_graph_pairs / k_graph_resolve, which spread it over many workgroups.
        if (MODE == YP_MODE_YOLOV3 && s.gseg != nullptr && K > kGraphMin && K <= (1 << 20) &&
            s.thr >= 0.05f && s.thr <= 1.0f) {
            float *s_redf = reinterpret_cast<float *>(&s_chunk);   // [kSegWaves][8] partial results
            GraphSeg *s_g = reinterpret_cast<GraphSeg *>(s_redf + 8 * kSegWaves);
            static_assert(sizeof(ChunkShared) >= 8 * kSegWaves * 4 + sizeof(GraphSeg) + 16, "scratch fits");
            int *s_hist = reinterpret_cast<int *>(s_lds), *s_cur = s_hist + kGraphBins;
            static_assert(2 * kGraphBins * 4 <= kRawBytes + kSmallBytes, "histogram and cursors fit");
            const float inf = __builtin_inff();
            float xlo = inf, xhi = -inf, ylo = inf, yhi = -inf, mlo = inf, mhi = -inf;
            bool ok = true;
            for (int i = tid; i < K; i += kSegThreads) {
                const float4 c4 = gbox[i];
                const float w = c4.z - c4.x + 1.0f, h = c4.w - c4.y + 1.0f;
                const float size = w > h ? w : h;
                const float cx = 0.5f * (c4.x + c4.z), cy = 0.5f * (c4.y + c4.w);
                // (comparisons that a NaN fails; ordered corners make every width and height >= 1,
                // which with a finite area is what the packed pair test of the graph kernels assumes)
                ok = ok && (c4.z >= c4.x) && (c4.w >= c4.y) && (w * h < 1e30f) && (size < 1e30f) &&
                     (__builtin_fabsf(cx) < 1e30f) && (__builtin_fabsf(cy) < 1e30f);
                xlo = vmin(xlo, cx); xhi = vmax(xhi, cx);
                ylo = vmin(ylo, cy); yhi = vmax(yhi, cy);
                mlo = vmin(mlo, size); mhi = vmax(mhi, size);
            }
            xlo = wave_min_f(xlo); xhi = wave_max_f(xhi);
            ylo = wave_min_f(ylo); yhi = wave_max_f(yhi);
            mlo = wave_min_f(mlo); mhi = wave_max_f(mhi);
            const bool wave_ok = __ballot(ok) == ~0ull;
            if (lane == 0) {
                float *r = s_redf + 8 * wave;
                r[0] = xlo; r[1] = xhi; r[2] = ylo; r[3] = yhi; r[4] = mlo; r[5] = mhi;
                r[6] = wave_ok ? 1.0f : 0.0f;
            }
            if (tid < kGraphClasses) s_cur[tid] = 0;
            __syncthreads();
            bool all_ok = true;
            xlo = inf; xhi = -inf; ylo = inf; yhi = -inf; mlo = inf; mhi = -inf;
            for (int w = 0; w < kSegWaves; ++w) {
                const float *r = s_redf + 8 * w;
                xlo = vmin(xlo, r[0]); xhi = vmax(xhi, r[1]);
                ylo = vmin(ylo, r[2]); yhi = vmax(yhi, r[3]);
                mlo = vmin(mlo, r[4]); mhi = vmax(mhi, r[5]);
                all_ok = all_ok && r[6] != 0.0f;
            }
            if (all_ok) {     // (uniform over the workgroup)
                const int e_min = (int)((__float_as_uint(mlo) >> 23) & 255u);
                for (int i = tid; i < K; i += kSegThreads)
                    atomicAdd(&s_cur[graph_class(graph_box_size<MODE>(gbox[i]), e_min)], 1);
                __syncthreads();
                if (tid == 0) {
                    GraphSeg g;
                    g.K = K;
                    g.table = atomicAdd(&s.ctrl[1], 1);
                    atomicAdd(&s.ctrl[2], (K + 63) >> 6);
                    g.e_min = e_min;
                    int d = 0;
                    while ((float)(1 << d) < 1.001f / s.thr) ++d;
                    g.d = d;
                    g.x0 = xlo;
                    g.y0 = ylo;
                    g.margin = 2.0f + 4e-7f * (__builtin_fabsf(xlo) + __builtin_fabsf(xhi) +
                                               __builtin_fabsf(ylo) + __builtin_fabsf(yhi) + mhi);
                    g.shrink = 0.9f * s.thr < 0.5f ? 0.9f * s.thr : 0.5f;
                    g.cols = graph_cols(K);
                    const float xr = xhi - xlo, yr = yhi - ylo;
                    int base = 0;
                    for (int c = 0; c < kGraphClasses; ++c) {
                        const int n_c = s_cur[c];
                        g.nx[c] = g.ny[c] = 0;
                        g.base[c] = base;
                        g.invx[c] = g.invy[c] = 0.0f;
                        g.half[c] = 0.0f;
                        if (n_c == 0) continue;
                        // sizes of class c < last have biased exponent e_min + c, i.e. lie below
                        // 2^(e_min + c + 1 - 127); the last class takes everything above its start
                        float ub = __uint_as_float((unsigned)(e_min + c + 1 < 254 ? e_min + c + 1 : 254) << 23);
                        if (c == kGraphClasses - 1) ub = vmax(ub, mhi * 1.0001f);
                        // (half the size of the largest core of the class, plus the half pixel between
                        // a box's centre and the centre of the interval the IoU counts)
                        g.half[c] = (1.0f - 2.0f * g.shrink) * 0.5f * ub + 1.0f;
                        // The class gets a share of the bins in proportion to its boxes.  The
                        // consecutive index positions of one batch of columns (32 or 64) are a run of
                        // cells along x, so the cells are made as tall as such a run is long -- about
                        // sqrt(columns / density) -- and as narrow as the share allows: the batch then
                        // covers a roughly square patch, which keeps its reach small.
                        const int budget = 8 + (int)((long long)(kGraphBins - 8 * kGraphClasses) * n_c / K);
                        int ny = 1;
                        if (yr > 0.0f && xr > 0.0f) {
                            const float h = __builtin_sqrtf((float)g.cols * xr * (yr / (float)n_c));
                            const float fy = h > 0.0f ? yr / h : 1.0f;
                            ny = fy < 1.0f ? 1 : (fy > (float)budget ? budget : (int)fy);
                        } else if (yr > 0.0f) {
                            ny = budget;
                        }
                        int nx = xr > 0.0f ? budget / ny : 1;
                        nx = nx < 1 ? 1 : nx;
                        g.nx[c] = nx;
                        g.ny[c] = ny;
                        g.invx[c] = xr > 0.0f ? (float)nx / xr : 0.0f;
                        g.invy[c] = yr > 0.0f ? (float)ny / yr : 0.0f;
                        if (!(g.invx[c] < 3e38f)) g.invx[c] = 0.0f, g.nx[c] = 1;   // (degenerate range)
                        if (!(g.invy[c] < 3e38f)) g.invy[c] = 0.0f, g.ny[c] = 1;
                        base += g.nx[c] * g.ny[c];
                    }
                    g.n_bins = base;
                    int pos = 0, nb = 0;
                    for (int c = 0; c < kGraphClasses; ++c) {
                        g.cpos[c] = pos;
                        g.cbat[c] = nb;
                        pos += s_cur[c];
                        nb += (s_cur[c] + g.cols - 1) / g.cols;
                    }
                    g.cpos[kGraphClasses] = pos;
                    g.cbat[kGraphClasses] = nb;
                    g.n_batches = nb;
                    *s_g = g;
                }
                __syncthreads();
                const GraphSeg &g = *s_g;
                for (int i = tid; i < kGraphBins; i += kSegThreads) s_hist[i] = 0;
                __syncthreads();
                auto bin_of = [&](const float4 &c4) {
                    const int c = graph_class(graph_box_size<MODE>(c4), g.e_min);
                    const int ix = graph_cell(0.5f * (c4.x + c4.z), g.x0, g.invx[c], g.nx[c]);
                    const int iy = graph_cell(0.5f * (c4.y + c4.w), g.y0, g.invy[c], g.ny[c]);
                    return g.base[c] + iy * g.nx[c] + ((iy & 1) ? g.nx[c] - 1 - ix : ix);
                };
                for (int i = tid; i < K; i += kSegThreads) atomicAdd(&s_hist[bin_of(gbox[i])], 1);
                __syncthreads();
                {   // exclusive scan of the kGraphBins counts: 8 consecutive bins per thread
                    constexpr int kPer = kGraphBins / kSegThreads;
                    int loc[kPer], sum = 0;
#pragma unroll
                    for (int k = 0; k < kPer; ++k) {
                        loc[k] = sum;
                        sum += s_hist[tid * kPer + k];
                    }
                    const int incl = wave_inclusive_scan(sum, lane);
                    if (lane == 63) s_scan[wave] = incl;
                    __syncthreads();
                    int woff = 0;
                    for (int w = 0; w < kSegWaves; ++w) woff += w < wave ? s_scan[w] : 0;
                    int *table = s.bin_tables + (size_t)g.table * (kGraphBins + 1);
#pragma unroll
                    for (int k = 0; k < kPer; ++k) {
                        const int o = woff + incl - sum + loc[k];
                        s_cur[tid * kPer + k] = o;
                        table[tid * kPer + k] = o;
                    }
                    if (tid == 0) table[kGraphBins] = K;
                }
                __syncthreads();
                float4 *xraw = reinterpret_cast<float4 *>(gk + 2 * (size_t)region);
                int *xidx = s.xidx + reg;
                int *gcnt = s.gcnt + reg, *gedge = s.gedge + reg * kGraphEdges;
                for (int i = tid; i < K; i += kSegThreads) {
                    const float4 c4 = gbox[i];
                    const int pos = atomicAdd(&s_cur[bin_of(c4)], 1);
                    xidx[pos] = i;
                    xraw[pos] = c4;
                    gcnt[pos] = 0;
                    gedge[(size_t)pos * kGraphEdges] = 0x7fffffff;
                }
                if (tid == 0) {
                    s.glist[g.table] = seg;
                    s.gseg[seg] = g;
                    s.seg_kept[seg] = 0;
                    s.seg_pairs[seg] = 0u;
                    s.seg_ties[seg] = 0u;
                    s.ctrl[0] = 1;
                }
                STAMP(5);   // (diagnostic build: the index build shows as the NMS phase)
                STAMP(6);
                return;
            }
            __syncthreads();   // not eligible: the scratch goes back to the chunked path below
        }
        // ---- x-order of the candidates: counting sort by (stripe, width class, position inside
        // the stripe) of the box centre along the pruning axis into 8 x 32 buckets (LDS
        // histogram; the order inside a bucket is whatever the atomics give, which only changes
        // which lane owns which column, never a verdict).  Inside each of the 8 stripes -- about
        // one per wave -- narrow, medium and wide boxes form runs of their own: a batch of 64
        // neighbours then spans a band about as wide as its own boxes instead of as wide as the
        // widest box around, and far fewer waiting rows reach it.
        // The sort keys are no longer needed, so their global region becomes the x-ordered
        // copy of the boxes and s_raw the histogram.
        constexpr int kXB = 256;
        static_assert(kXB <= kSegThreads && (2 * kXB + 4 * kSegWaves) * 4 <= kRawBytes, "histogram fits");
        int *s_hist = reinterpret_cast<int *>(s_raw);
        int *s_cur = s_hist + kXB;
        float *s_red = reinterpret_cast<float *>(s_cur + kXB);       // [2 * kSegWaves] min, max
        int *s_pos = reinterpret_cast<int *>(s_red + 2 * kSegWaves); // [kSegWaves] all areas > 0
        float4 *xraw = reinterpret_cast<float4 *>(gk + 2 * (size_t)region);
        int *xidx = s.xidx + reg;
        float width = 0.0f;   // of the box axis_key looked at last, along the pruning axis
        auto axis_key = [&](const float4 &raw, bool &positive) {
            const YpBox bx = {raw.x, raw.y, raw.z, raw.w};
            const YpPrep pp = yp_prep<MODE>(bx);
            positive = pp.area > 0.0f;
            width = MODE == YP_MODE_YOLOV3 ? pp.hi0 - pp.lo0 : pp.hi1 - pp.lo1;
            return MODE == YP_MODE_YOLOV3 ? pp.lo0 + pp.hi0 : pp.lo1 + pp.hi1;
        };
        {
            float kmin = __builtin_inff(), kmax = -__builtin_inff();
            bool allpos = true;
            for (int i = tid; i < K; i += kSegThreads) {
                bool positive;
                const float key = axis_key(gbox[i], positive);
                kmin = vmin(kmin, key);
                kmax = vmax(kmax, key);
                allpos = allpos && positive;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                kmin = vmin(kmin, __shfl_xor(kmin, off, 64));
                kmax = vmax(kmax, __shfl_xor(kmax, off, 64));
            }
            const u64 posm = __ballot(allpos);
            if (lane == 0) {
                s_red[wave] = kmin;
                s_red[kSegWaves + wave] = kmax;
                s_pos[wave] = posm == ~0ull ? 1 : 0;
            }
            if (tid < kXB) s_hist[tid] = 0;
        }
        __syncthreads();
        float kmin = s_red[0], kmax = s_red[kSegWaves];
        bool allpos = s_pos[0] != 0;
#pragma unroll
        for (int w = 1; w < kSegWaves; ++w) {
            kmin = vmin(kmin, s_red[w]);
            kmax = vmax(kmax, s_red[kSegWaves + w]);
            allpos = allpos && s_pos[w] != 0;
        }
        const float range = kmax - kmin;   // of lo + hi, i.e. twice the spread of the centres
        constexpr int kStripes = kSegWaves, kPerStripe = kXB / kStripes;
        static_assert(kPerStripe == 32, "bins per stripe: 14 narrow + 10 medium + 8 wide");
        const float scale = (range > 0.0f && range < 3.0e38f) ? (float)kStripes / range : 0.0f;
        const float w_narrow = range * (1.0f / 96.0f), w_medium = range * (1.0f / 24.0f);
        auto bucket = [&](float key) {     // uses the width axis_key left behind
            float t = (key - kmin) * scale;
            t = t > 0.0f ? t : 0.0f;                       // also turns a NaN into stripe 0
            t = t < (float)kStripes ? t : (float)kStripes;
            int stripe = (int)t;
            stripe = stripe < kStripes ? stripe : kStripes - 1;
            float f = t - (float)stripe;                   // position inside the stripe, [0, 1]
            f = f < 1.0f ? f : 0.999f;
            const bool narrow = width <= w_narrow, medium = width <= w_medium;   // NaN: wide
            const int first = narrow ? 0 : (medium ? 14 : 24), bins = narrow ? 14 : (medium ? 10 : 8);
            return stripe * kPerStripe + first + (int)(f * (float)bins);
        };
        for (int i = tid; i < K; i += kSegThreads) {
            bool positive;
            atomicAdd(&s_hist[bucket(axis_key(gbox[i], positive))], 1);
        }
        __syncthreads();
        if (tid < kXB) {   // exclusive scan of the 256 counts by the first four waves
            const int v = s_hist[tid];
            const int incl = wave_inclusive_scan(v, lane);
            if (lane == 63) s_scan[wave] = incl;
            s_cur[tid] = incl - v;
        }
        __syncthreads();
        if (tid < kXB) {
            int woff = 0;
            for (int w = 0; w < kXB / 64; ++w) woff += w < wave ? s_scan[w] : 0;
            s_cur[tid] += woff;
        }
        __syncthreads();
        for (int i = tid; i < K; i += kSegThreads) {
            bool positive;
            const float4 raw = gbox[i];
            const int pos = atomicAdd(&s_cur[bucket(axis_key(raw, positive))], 1);
            xidx[pos] = i;
            xraw[pos] = raw;
        }
        __syncthreads();
        const YpThr thr = yp_make_thr(s.thr);
        const bool prune = thr.fast && (MODE == YP_MODE_REFERENCE || allpos);
        nkept = nms_chunks<MODE>(gbox, grem, K, thr, max_out, s_chunk, xraw, xidx, prune, pairs, ties,
#ifdef YP_STAMPS
                                 s.dbg ? s.dbg + (size_t)vblk * 32 : nullptr,
#else
                                 nullptr,
#endif
                                 [&](int m, int sp, const float4 &bx) { emit_kept(gslot, m, sp, bx); });
    }
    (void)sslot_final;

    STAMP(5);
    pairs = (unsigned int)wave_sum((int)pairs);
    ties = (unsigned int)wave_sum((int)ties);
    if (lane == 0 && pairs) atomicAdd(&s_pairs, pairs);
    if (lane == 0 && ties) atomicAdd(&s_ties, ties);
    __syncthreads();
    if (tid == 0) {
        s.seg_kept[seg] = nkept;
        s.seg_pairs[seg] = s_pairs;
        s.seg_ties[seg] = s_ties;
        if (MODE == YP_MODE_REFERENCE) {
            s.out_counts[b] = nkept;
            s.surv_counts[b] = K;
        }
    }
    STAMP(6);
}

// One workgroup per segment.
template <int MODE>
__global__ __launch_bounds__(kSegThreads)
__attribute__((amdgpu_waves_per_eu(6, 6)))   // 80 vector registers: three workgroups per CU
void k_segment_nms(SegArgs s)
{
    segment_nms_body<MODE>(s, (int)blockIdx.x, (int)gridDim.x);
}


// --------------------------------------------------------------------- final row gather
constexpr int kGatherThreads = 256;
__global__ __launch_bounds__(kGatherThreads) void k_gather_rows(GatherArgs g)
{
    __shared__ int s_part[kGatherThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int seg = blockIdx.x;
    // every set (independent batch) has its own output rows and count: the prefix runs over the
    // earlier segments of this segment's set only
    const int set = seg / g.spset, seg0 = set * g.spset;
    int part = 0;
    for (int i = seg0 + tid; i < seg; i += kGatherThreads) part += g.seg_kept[i];
    part = wave_sum(part);
    int prefix = __shfl(part, 0, 64);   // wave_sum leaves the total in lane 0
    if (kGatherThreads > 64) {
        if (lane == 0) s_part[wave] = part;
        __syncthreads();
        prefix = 0;
#pragma unroll
        for (int w = 0; w < kGatherThreads / 64; ++w) prefix += s_part[w];
    }
    const int nk = g.seg_kept[seg];
    if (tid == 0) {
        if (seg == seg0 + g.spset - 1) g.out_count[set] = prefix + nk;
        if (g.stats) {   // (n_sets, 4, images per set, C)
            int *st = g.stats + (size_t)set * 4 * g.spset + (seg - seg0);
            st[0] = g.seg_cand[seg];
            st[g.spset] = nk;
            st[2 * g.spset] = (int)g.seg_pairs[seg];
            st[3 * g.spset] = (int)g.seg_ties[seg];
        }
    }
    const int b = seg / g.C, c = seg - b * g.C;
    const int b_in_set = b - set * (g.spset / g.C);
    const size_t ibase = (size_t)b * g.spi;
    const size_t reg = ibase + g.seg_region[seg];
    float *rows_out = g.out_rows + (size_t)set * g.set_stride;
    for (int k = tid; k < nk; k += kGatherThreads) {
        const long long d = (long long)prefix + k;
        if (d >= g.cap) break;
        const int slot = g.kept_slot[reg + k];
        float4 bx = g.cand_box[ibase + slot];
        if (g.rescale) {
            bx.x = vmin(vmax((bx.x - g.pad_x) / g.scale, 0.0f), g.im_w);
            bx.y = vmin(vmax((bx.y - g.pad_y) / g.scale, 0.0f), g.im_h);
            bx.z = vmin(vmax((bx.z - g.pad_x) / g.scale, 0.0f), g.im_w);
            bx.w = vmin(vmax((bx.w - g.pad_y) / g.scale, 0.0f), g.im_h);
        }
        float4 *o = reinterpret_cast<float4 *>(rows_out + d * 8);
        o[0] = make_float4((float)(b_in_set + g.batch_offset), bx.x, bx.y, bx.z);
        o[1] = make_float4(bx.w, g.cand_key[ibase + slot], g.cand_aux[ibase + slot], (float)c);
    }
}

// The same with one workgroup per IMAGE (up to kGatherImgClasses classes): the rows of earlier
// images of the set are summed once per image instead of once per segment, and the image's rows of
// all classes are written by one flat loop, so a few hundred rows keep 256 threads busy.
constexpr int kGatherImgClasses = 64;
constexpr int kGatherImgMinImages = 128;
__global__ __launch_bounds__(kGatherThreads) void k_gather_image(GatherArgs g)
{
    __shared__ int s_part[kGatherThreads / 64];
    __shared__ int s_pre[kGatherImgClasses + 1];   // kept rows of the image before class c
    __shared__ int s_reg[kGatherImgClasses];       // where class c's kept slots start in the image's region
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x, C = g.C;
    const int bps = g.spset / C;                   // images per set
    const int set = b / bps, b_in_set = b - set * bps;
    const int seg0 = set * g.spset, segb = b * C;
    int part = 0;
    for (int i = seg0 + tid; i < segb; i += kGatherThreads) part += g.seg_kept[i];
    part = wave_sum(part);
    if (lane == 0) s_part[wave] = part;
    if (wave == 0) {
        const int nk = lane < C ? g.seg_kept[segb + lane] : 0;
        const int incl = wave_inclusive_scan(nk, lane);
        if (lane < C) {
            s_pre[lane + 1] = incl;
            s_reg[lane] = g.seg_region[segb + lane];
        }
        if (lane == 0) s_pre[0] = 0;
        if (g.stats && lane < C) {   // (n_sets, 4, images per set, C)
            int *st = g.stats + (size_t)set * 4 * g.spset + (segb - seg0) + lane;
            st[0] = g.seg_cand[segb + lane];
            st[g.spset] = nk;
            st[2 * g.spset] = (int)g.seg_pairs[segb + lane];
            st[3 * g.spset] = (int)g.seg_ties[segb + lane];
        }
    }
    __syncthreads();
    int prefix = 0;
#pragma unroll
    for (int w = 0; w < kGatherThreads / 64; ++w) prefix += s_part[w];
    const int total = s_pre[C];
    if (tid == 0 && b_in_set == bps - 1) g.out_count[set] = prefix + total;
    const size_t ibase = (size_t)b * g.spi;
    float *rows_out = g.out_rows + (size_t)set * g.set_stride;
    for (int k = tid; k < total; k += kGatherThreads) {
        const long long d = (long long)prefix + k;
        if (d >= g.cap) break;
        int c = 0, hi = C;   // the class whose rows hold position k: s_pre[c] <= k < s_pre[c + 1]
        while (hi - c > 1) {
            const int mid = (c + hi) >> 1;
            if (k >= s_pre[mid]) c = mid;
            else hi = mid;
        }
        const int slot = g.kept_slot[ibase + s_reg[c] + (k - s_pre[c])];
        float4 bx = g.cand_box[ibase + slot];
        if (g.rescale) {
            bx.x = vmin(vmax((bx.x - g.pad_x) / g.scale, 0.0f), g.im_w);
            bx.y = vmin(vmax((bx.y - g.pad_y) / g.scale, 0.0f), g.im_h);
            bx.z = vmin(vmax((bx.z - g.pad_x) / g.scale, 0.0f), g.im_w);
            bx.w = vmin(vmax((bx.w - g.pad_y) / g.scale, 0.0f), g.im_h);
        }
        float4 *o = reinterpret_cast<float4 *>(rows_out + d * 8);
        o[0] = make_float4((float)(b_in_set + g.batch_offset), bx.x, bx.y, bx.z);
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

#include "yp_epilogue.h"

// Reference predict_transform (detect_net.py:152-183): thread <-> row.  The raw rows come in,
// and the class probabilities go out, through the tile staging of yp_tile.h; a record is
// evaluated by yp_decode_record_ref (yp_epilogue.h).
template <bool STAGED, int CS>   // CS as in k_front_grid_ref
__global__ void k_decode_grid_ref(const float *pred, int H, int W, int A, int C_arg, AnchorTable at,
                                  float *center, float *wh, float *obj, float *cls, long long total)
{
    const int C = CS ? CS : C_arg;
    extern __shared__ float s_tile[];
    const long long tile0 = (long long)blockIdx.x * kTileRows;
    const long long o = tile0 + threadIdx.x;
    const int N = H * W * A, V = 5 + C;
    if (STAGED) yp_tile_load(pred + tile0 * V, (total - tile0) * V, V, s_tile);
    float probs[STAGED ? kTileMaxWidth : 1];
    if (o < total) {
        const int n = (int)(o % N);
        const float *t = STAGED ? s_tile + threadIdx.x * V : pred + o * V;
        float2 ce, sz;
        float ob;
        if (STAGED) {
            yp_decode_record_ref<CS>(t, n, H, W, A, C, at, ce, sz, ob, probs);
        } else {
            // records too wide for the staging buffer: the probabilities go straight to memory
            yp_decode_record_ref<0>(t, n, H, W, A, C, at, ce, sz, ob, cls + o * C);
        }
        reinterpret_cast<float2 *>(center)[o] = ce;
        reinterpret_cast<float2 *>(wh)[o] = sz;
        obj[o] = ob;
    }
    if (STAGED) {
        __syncthreads();   // every thread has read its raw row: the buffer now takes the results
        if (o < total) {
#pragma unroll
            for (int j = 0; j < kTileMaxWidth; ++j)
                if (j < C) s_tile[threadIdx.x * C + j] = probs[j];
        }
        yp_tile_store(cls + tile0 * C, (total - tile0) * C, C, s_tile);
    }
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
unsigned long long *g_dbg = nullptr;  // set only through yp_debug_set_stamps (diagnostic build)
unsigned long long *g_dbg_front = nullptr;

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
    int *xidx;
    int *seg_kept;
    int *seg_cand;
    int *seg_region;
    unsigned int *seg_pairs;
    unsigned int *seg_ties;
    // graph path (YOLOv3 mode only)
    int *gcnt;
    int *gedge;
    GraphSeg *gseg;
    int *bin_tables;
    int *glist;
    int *ctrl;
    int *img_flag;
};

template <class T>
size_t carve_one(char *base, size_t off, size_t count, T **out)
{
    off = (off + kAlign - 1) / kAlign * kAlign;
    if (base) *out = reinterpret_cast<T *>(base + off);
    return off + count * sizeof(T);
}

// Lays the scratch arrays out inside `base` (or just sizes them when base is null).
size_t carve(void *base_v, size_t slots, size_t tiles, size_t n_seg, Workspace *w, bool graph = true)
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
    off = carve_one(base, off, slots, &w->xidx);
    off = carve_one(base, off, n_seg, &w->seg_kept);
    off = carve_one(base, off, n_seg, &w->seg_cand);
    off = carve_one(base, off, n_seg, &w->seg_region);
    off = carve_one(base, off, n_seg, &w->seg_pairs);
    off = carve_one(base, off, n_seg, &w->seg_ties);
    off = carve_one(base, off, 16, &w->ctrl);
    off = carve_one(base, off, n_seg, &w->img_flag);   // (one per image; n_seg >= images)
    w->gcnt = nullptr;
    w->gedge = nullptr;
    w->gseg = nullptr;
    w->bin_tables = nullptr;
    w->glist = nullptr;
    if (graph) {
        // every graph segment has more than kGraphMin candidates of its own, which bounds how
        // many bin tables a launch can hand out
        const size_t n_tables = n_seg < slots / kGraphMin + 1 ? n_seg : slots / kGraphMin + 1;
        off = carve_one(base, off, slots, &w->gcnt);
        off = carve_one(base, off, slots * kGraphEdges, &w->gedge);
        off = carve_one(base, off, n_seg, &w->gseg);
        off = carve_one(base, off, n_tables * (kGraphBins + 1), &w->bin_tables);
        off = carve_one(base, off, n_tables, &w->glist);
    }
    return (off + kAlign - 1) / kAlign * kAlign;
}

bool params_ok(const YpParams *p, int mode)
{
    if (!p || p->mode != mode || p->batch <= 0 || p->batch > 65535) return false;
    if (p->num_classes <= 0 || p->num_classes > 4096) return false;
    if (!(p->nms_thr == p->nms_thr) || !(p->conf_thr == p->conf_thr)) return false;
#ifdef YP_DIAG
    if ((p->flags & ~(YP_FLAG_DENSE_NMS | YP_FLAG_RESCALE | 0x700)) != 0) return false;   // bits 8-10: stage mask
#else
    if ((p->flags & ~(YP_FLAG_DENSE_NMS | YP_FLAG_RESCALE)) != 0) return false;
#endif
    if (p->flags & YP_FLAG_RESCALE) {
        // only the YOLOv3 entries that see the head geometry rescale; the frame must be a real one
        if (mode != YP_MODE_YOLOV3 || !(p->im_h > 0.0f) || !(p->im_w > 0.0f) || !(p->im_h < 1e9f) ||
            !(p->im_w < 1e9f))
            return false;
    }
    return true;
}

// Geometry of a set of YOLOv3 heads: tiles of cells_per_tile cells, slots per tile =
// cells_per_tile * max anchors.
int heads_geometry(const YpHead *heads, int n_heads, int C, int cells_per_tile, HeadsArgs *ha,
                   int *amax_out, int n_sets = 1)
{
    if (!heads || n_heads < 1 || n_heads > YP_MAX_HEADS) return YP_ERR_ARG;
    if (n_sets < 1 || n_sets > YP_MAX_SETS) return YP_ERR_UNSUPPORTED;
    // sets are set-major in `heads`; every set must repeat the geometry of set 0
    for (int s = 0; s < YP_MAX_SETS; ++s)
        for (int i = 0; i < YP_MAX_HEADS; ++i) {
            const bool used = s < n_sets && i < n_heads;
            const YpHead &h = heads[used ? s * n_heads + i : 0], &h0 = heads[used ? i : 0];
            if (used && (!h.data || h.grid_h != h0.grid_h || h.grid_w != h0.grid_w ||
                         h.num_anchors != h0.num_anchors || h.stride_x != h0.stride_x ||
                         h.stride_y != h0.stride_y))
                return YP_ERR_ARG;
            if (used)
                for (int k = 0; k < 2 * YP_MAX_ANCHORS && k < 2 * h0.num_anchors; ++k)
                    if (h.anchors[k] != h0.anchors[k]) return YP_ERR_ARG;
            ha->data[s][i] = h.data;
        }
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
        tiles += ((long long)h.grid_h * h.grid_w + cells_per_tile - 1) / cells_per_tile;
        rows += (long long)h.grid_h * h.grid_w * h.num_anchors;
        amax = h.num_anchors > amax ? h.num_anchors : amax;
    }
    for (int i = n_heads; i < YP_MAX_HEADS; ++i) ha->h[i] = ha->h[0];
    if (tiles > (1ll << 22) || rows > (1ll << 28)) return YP_ERR_UNSUPPORTED;
    ha->dbg = nullptr;
    ha->bps = 0;   // set by the caller that launches a front kernel
    ha->n_heads = n_heads;
    ha->tpi = (int)tiles;
    ha->C = C;
    ha->N = (int)rows;
    *amax_out = amax;
    return YP_OK;
}

// Guard band around logit(conf) outside which the confidence filter sigmoid(t) > conf is
// decided by comparing t: sigmoid(t_hi) exceeds conf, and sigmoid(t_lo) falls short of it, by
// orders of magnitude more than the error of the computed sigmoid.  For conf outside
// [1e-6, 0.99] the band is infinite, i.e. every objectness goes through the sigmoid.
void logit_band(float conf, float *t_lo, float *t_hi)
{
    *t_lo = -__builtin_inff();
    *t_hi = __builtin_inff();
    if (conf >= 1e-6f && conf <= 0.99f) {
        const double c = (double)conf;
        const double l = __builtin_log(c / (1.0 - c));
        const double d = 1e-3 * (__builtin_fabs(l) > 1.0 ? __builtin_fabs(l) : 1.0);
        *t_lo = (float)(l - d);
        *t_hi = (float)(l + d);
    }
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
    s.xidx = w.xidx;
    s.seg_kept = w.seg_kept;
    s.seg_cand = w.seg_cand;
    s.seg_region = w.seg_region;
    s.seg_pairs = w.seg_pairs;
    s.seg_ties = w.seg_ties;
    s.gseg = w.gseg;
    s.bin_tables = w.bin_tables;
    s.glist = w.glist;
    s.gcnt = w.gcnt;
    s.gedge = w.gedge;
    s.ctrl = w.ctrl;
    s.img_flag = nullptr;
    s.dbg = g_dbg;
    return s;
}

// Optional per-kernel timing: while enabled, every kernel of a detect call is launched with
// a start and a stop event of its own (hipExtLaunchKernelGGL), which carry the begin and end
// time stamps of that dispatch; their difference is the kernel's execution time on the
// device, without the host-side gaps a pair of hipEventRecord calls would include.
struct ProfileState {
    bool on = false;
    int max_calls = 0;
    int calls = 0;
    hipEvent_t *ev = nullptr;        // kProfEvents per call: (start, stop) x (front, nms, gather,
                                     // graph pairs, graph resolve)
    unsigned char *launched = nullptr;  // per call: bit k set when kernel k was launched
};
constexpr int kProfKernels = 7, kProfEvents = 2 * kProfKernels;   // ([5] k_image_nms, [6] the second k_graph_pairs)
constexpr int kProfReported = 5;   // yp_profile_stop_ex: k_image_nms is added to the segment NMS time, the two pair kernels are one
ProfileState g_prof;

inline bool prof_active() { return g_prof.on && g_prof.calls < g_prof.max_calls; }

inline hipEvent_t prof_ev(int k, int end)
{
    g_prof.launched[g_prof.calls] |= (unsigned char)(1u << k);
    return g_prof.ev[kProfEvents * g_prof.calls + 2 * k + end];
}

inline void prof_next()
{
    if (prof_active()) ++g_prof.calls;
}

// k: 0 front, 1 segment NMS, 2 gather, 3 graph pairs, 4 graph resolve, 5 image NMS.
#define YP_LAUNCH(k, kernel, grid, block, lds, st, ...)                                          \
    do {                                                                                         \
        if (prof_active())                                                                       \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, st, prof_ev(k, 0), prof_ev(k, 1), 0, \
                                  __VA_ARGS__);                                                  \
        else                                                                                     \
            hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                       \
    } while (0)

// Smallest launch (images) that takes the per-image NMS kernel: kImgMinImages unless the
// environment says otherwise (read at every call: a test may change it between calls).
int image_path_min_images()
{
    const char *e = getenv("YP_IMG_MIN_IMAGES");
    if (e == nullptr || *e == 0) return kImgMinImages;
    char *end = nullptr;
    const long v = strtol(e, &end, 10);
    return (end == e || v < 1) ? kImgMinImages : (v > 0x7fffffffl ? 0x7fffffff : (int)v);
}

// Segment NMS + final gather of the YOLOv3 path (shared by the heads and rows entries).
int finish_v3(const Workspace &w, const YpParams *p, int n_sets, int spi, int tpi, int ts,
              float *out_rows, long long cap, int32_t *out_count, int32_t *stats, hipStream_t st,
              float inp_w = 0.0f, float inp_h = 0.0f)
{
    const int n_seg = n_sets * p->batch * p->num_classes;
#ifdef YP_DIAG
    const int stages = (p->flags >> 8) & 7 ? (p->flags >> 8) & 7 : 7;   // diagnostic build: a subset of the stages
#else
    const int stages = 7;
#endif
    SegArgs s = make_seg_args(w, p, spi, tpi, ts, 1);
    s.B = n_sets * p->batch;
    s.max_out = p->max_out;
    // Large segments go to the graph kernels only when the caller asks for it: the two extra
    // launches cost a few microseconds per call even when no segment is large.
    const bool dense = (p->flags & YP_FLAG_DENSE_NMS) != 0 && w.gseg != nullptr && spi > kGraphMin;
    if (!dense) s.gseg = nullptr;
    // Images of moderate size are finished by one workgroup each (k_image_nms); that kernel flags
    // the images it cannot take, and k_segment_nms, launched behind it, only works on those.
    const bool per_image = p->num_classes <= kImgMaxClasses && tpi <= kImgMaxTiles &&
                           n_sets * p->batch >= image_path_min_images() && yp_make_thr(p->nms_thr).fast;
    if (per_image) s.img_flag = w.img_flag;
    if ((stages & 2) && per_image) {
        const dim3 ig(n_sets * p->batch);
        if (tpi <= kImgSmallTiles)
            YP_LAUNCH(5, (k_image_nms<YP_MODE_YOLOV3, kImgThreads, kImgCapSmall, kImgPoolSmall>), ig,
                      dim3(kImgThreads), 0, st, s);
        else
            YP_LAUNCH(5, (k_image_nms<YP_MODE_YOLOV3, kImgThreads, kImgCapLarge, kImgPoolLarge>), ig,
                      dim3(kImgThreads), 0, st, s);
    }
    if (stages & 2)
        YP_LAUNCH(1, k_segment_nms<YP_MODE_YOLOV3>, dim3(n_seg), dim3(kSegThreads), 0, st, s);
    if ((stages & 2) && dense) {
        // segments the NMS kernel handed to the graph path (none: both kernels leave at once)
        GraphArgs ga = {};
        ga.per_class = 1;
        ga.C = p->num_classes;
        ga.spi = spi;
        ga.max_out = p->max_out;
        ga.thr = p->nms_thr;
        ga.gseg = w.gseg;
        ga.bin_tables = w.bin_tables;
        ga.xraw = reinterpret_cast<const float4 *>(w.gkeys);
        ga.xidx = w.xidx;
        ga.status = w.removed;
        ga.gcnt = w.gcnt;
        ga.gedge = w.gedge;
        ga.sslot = w.sslot;
        ga.seg_region = w.seg_region;
        ga.glist = w.glist;
        ga.kept_slot = w.kept_slot;
        ga.seg_kept = w.seg_kept;
        ga.seg_pairs = w.seg_pairs;
        ga.seg_ties = w.seg_ties;
        ga.work = reinterpret_cast<int *>(w.sbox);
        ga.dbg = g_dbg ? g_dbg + (size_t)n_seg * 32 : nullptr;   // (diagnostic build: one row behind the segments')
        // pair tests: a fixed number of workgroups that share out the batches of all graph
        // segments; resolve: one workgroup per graph segment (those beyond their number leave)
        const long long slots_all = (long long)n_seg / p->num_classes * spi;
        long long n_tables = slots_all / kGraphMin + 1;
        n_tables = n_tables < n_seg ? n_tables : n_seg;
        long long n_wg = slots_all / 64 / 2 + 1;
        n_wg = n_wg > kGraphPairGrid ? kGraphPairGrid : n_wg;
        // (one launch per batch width: segments of fewer than kGraphWideMin candidates, then the larger
        // ones; a launch without segments of its kind ends after a look at the list)
        YP_LAUNCH(3, (k_graph_pairs<YP_MODE_YOLOV3, kGraphColsSmall>), dim3((unsigned)n_wg), dim3(kGraphThreads), 0,
                  st, ga, w.ctrl);
        if (spi >= kGraphWideMin)
            YP_LAUNCH(6, (k_graph_pairs<YP_MODE_YOLOV3, 64>), dim3((unsigned)n_wg), dim3(kGraphThreads), 0, st, ga,
                      w.ctrl);
        YP_LAUNCH(4, k_graph_resolve<YP_MODE_YOLOV3>, dim3((unsigned)n_tables), dim3(kSegThreads), 0, st, ga,
                  w.ctrl);
    }
    GatherArgs g = {};
    g.n_seg = n_seg;
    g.C = p->num_classes;
    g.spi = spi;
    g.batch_offset = p->batch_offset;
    g.spset = p->batch * p->num_classes;
    g.set_stride = cap * 8;
    g.seg_ties = w.seg_ties;
    if (p->flags & YP_FLAG_RESCALE) {
        // letterbox geometry of a frame im_w x im_h resized into the inp_w x inp_h network input
        // with its aspect ratio kept and centred: every quantity in binary32, in this order
        const float sx = inp_w / p->im_w, sy = inp_h / p->im_h;
        g.rescale = 1;
        g.scale = sx < sy ? sx : sy;
        g.pad_x = (inp_w - g.scale * p->im_w) / 2.0f;
        g.pad_y = (inp_h - g.scale * p->im_h) / 2.0f;
        g.im_w = p->im_w;
        g.im_h = p->im_h;
    }
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
    // one workgroup per image once there are enough images to fill the GPU that way (an image may
    // hold tens of thousands of rows), else one per segment
    const int gather_min = kGatherImgMinImages < image_path_min_images() ? kGatherImgMinImages : image_path_min_images();
    if ((stages & 4) && p->num_classes <= kGatherImgClasses && n_sets * p->batch >= gather_min)
        YP_LAUNCH(2, k_gather_image, dim3(n_sets * p->batch), dim3(kGatherThreads), 0, st, g);
    else if (stages & 4)
        YP_LAUNCH(2, k_gather_rows, dim3(n_seg), dim3(kGatherThreads), 0, st, g);
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
    YP_LAUNCH(1, k_segment_nms<YP_MODE_REFERENCE>, dim3(p->batch), dim3(kSegThreads), 0, st, s);
    prof_next();
    return (int)hipGetLastError();
}

}  // namespace

// ================================================================================= C ABI

extern "C" {

int yp_version(void) { return 200; }

const char *yp_strerror(int code)
{
    switch (code) {
    case YP_OK: return "ok";
    case YP_ERR_ARG: return "invalid argument";
    case YP_ERR_WORKSPACE: return "workspace missing, misaligned or too small";
    case YP_ERR_UNSUPPORTED: return "shape beyond the compiled limits";
    case YP_ERR_SCRATCH:
        return "problem needs more scratch than the library's static buffer "
               "(yp_noalloc_scratch_bytes); use yp_detect_yolov3 with a caller workspace";
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
    const int blocks = (int)((total + kTileRows - 1) / kTileRows);
    const size_t lds = yp_tile_lds_bytes(5 + num_classes);
    if (lds && num_classes == 5)
        hipLaunchKernelGGL((k_decode_grid_ref<true, 5>), dim3(blocks), dim3(kTileRows), lds,
                           (hipStream_t)stream, d_pred, grid_h, grid_w, num_anchors, num_classes, at,
                           d_center, d_wh, d_obj, d_cls, total);
    else if (lds)
        hipLaunchKernelGGL((k_decode_grid_ref<true, 0>), dim3(blocks), dim3(kTileRows), lds,
                           (hipStream_t)stream, d_pred, grid_h, grid_w, num_anchors, num_classes, at,
                           d_center, d_wh, d_obj, d_cls, total);
    else
        hipLaunchKernelGGL((k_decode_grid_ref<false, 0>), dim3(blocks), dim3(kTileRows), 0,
                           (hipStream_t)stream, d_pred, grid_h, grid_w, num_anchors, num_classes, at,
                           d_center, d_wh, d_obj, d_cls, total);
    return (int)hipGetLastError();
}

int yp_dense_relu_predict_transform_reference(const float *d_features, const float *d_weight,
                                              const float *d_bias, int batch, int features, int grid_h,
                                              int grid_w, int num_anchors, int num_classes,
                                              const float *anchors, float *d_raw, float *d_center,
                                              float *d_wh, float *d_obj, float *d_cls, void *stream)
{
    if (!d_features || !d_weight || !d_bias || !anchors || !d_center || !d_wh || !d_obj || !d_cls)
        return YP_ERR_ARG;
    if (batch <= 0 || features <= 0 || grid_h <= 0 || grid_w <= 0) return YP_ERR_ARG;
    if (num_anchors < 1 || num_anchors > YP_MAX_ANCHORS) return YP_ERR_UNSUPPORTED;
    if (num_classes != 5) return YP_ERR_UNSUPPORTED;   // (the reference's class count, params.py:31)
    AnchorTable at = {};
    for (int i = 0; i < 2 * num_anchors; ++i) at.anc[i] = anchors[i];
    const long long total = (long long)batch * grid_h * grid_w * num_anchors;
    if (total >= (1ll << 31)) return YP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((k_dense_relu_decode_ref<5>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, d_features, d_weight, d_bias, features, grid_h, grid_w,
                       num_anchors, at, d_raw, d_center, d_wh, d_obj, d_cls, total);
    return (int)hipGetLastError();
}

int yp_predict_transform_yolov3(const YpHead *heads, int n_heads, int batch, int num_classes,
                                float *d_rows, void *stream)
{
    if (!d_rows || batch <= 0 || batch > 65535 || num_classes <= 0 || num_classes > 4096)
        return YP_ERR_ARG;
    HeadsArgs ha;
    int amax = 0;
    const int rc = heads_geometry(heads, n_heads, num_classes, 64, &ha, &amax);
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
    if (heads_geometry(tmp, n_heads, num_classes, kFrontCells, &ha, &amax) != YP_OK) return 0;
    const long long spi = (long long)ha.tpi * kFrontCells * amax;
    if (!slots_fit(batch, spi)) return 0;
    return carve(nullptr, (size_t)batch * spi, (size_t)batch * ha.tpi,
                 (size_t)batch * num_classes, nullptr, true);
}

// Shared body of yp_detect_yolov3 (one set) and yp_detect_yolov3_multi.
static int detect_v3_sets(const YpHead *heads, int n_heads, int n_sets, const YpParams *params,
                          float *d_out_rows, int64_t out_cap, int32_t *d_out_count,
                          int32_t *d_stats, void *d_workspace, size_t workspace_bytes,
                          void *stream)
{
    if (!params_ok(params, YP_MODE_YOLOV3) || !d_out_rows || !d_out_count || out_cap < 0)
        return YP_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(d_out_rows) & 15) != 0) return YP_ERR_ARG;
    if (n_sets < 1 || n_sets > YP_MAX_SETS) return YP_ERR_UNSUPPORTED;
    HeadsArgs ha;
    int amax = 0;
    const int rc = heads_geometry(heads, n_heads, params->num_classes, kFrontCells, &ha, &amax, n_sets);
    if (rc != YP_OK) return rc;
    const long long images = (long long)n_sets * params->batch;
    if (images > 65535) return YP_ERR_UNSUPPORTED;
    const int ts = kFrontCells * amax;
    const long long spi = (long long)ha.tpi * ts;
    if (!slots_fit(images, spi)) return YP_ERR_UNSUPPORTED;
    const size_t slots = (size_t)images * spi, tiles = (size_t)images * ha.tpi;
    const size_t n_seg = (size_t)images * params->num_classes;
    Workspace w;
    const size_t need = carve(nullptr, slots, tiles, n_seg, nullptr, true);
    if (!workspace_ok(d_workspace, workspace_bytes, need)) return YP_ERR_WORKSPACE;
    carve(d_workspace, slots, tiles, n_seg, &w, true);
    CandBuf cb = {w.cand_box, w.cand_key, w.cand_aux, w.cand_cls, w.tile_count, ha.tpi, ts, w.ctrl};
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(ha.tpi, params->batch, n_sets), block(kFrontCells * amax);
    float t_lo, t_hi;
    logit_band(params->conf_thr, &t_lo, &t_hi);
    ha.dbg = g_dbg_front;
    ha.bps = params->batch;
    bool all_three = true;
    for (int i = 0; i < n_heads; ++i) all_three = all_three && heads[i].num_anchors == 3;
#ifdef YP_DIAG
    const bool run_front = !((params->flags >> 8) & 7) || ((params->flags >> 8) & 1);
#else
    const bool run_front = true;
#endif
    if (!run_front) {
        // diagnostic build, stage mask without the front stage: the workspace keeps the
        // previous candidates
    } else if (params->num_classes == 5 && all_three)
        YP_LAUNCH(0, (k_front_heads_v3_wave<5, 3>), grid, dim3(64),
                  (size_t)kStageWin * ((6 + 5 + 3) & ~3) * sizeof(float), st, ha, cb, params->conf_thr, t_lo,
                  t_hi);
    else if (params->num_classes == 5)
        YP_LAUNCH(0, k_front_heads_v3<5>, grid, block, (size_t)ts * (6 + 5) * sizeof(float), st, ha, cb,
                  params->conf_thr, t_lo, t_hi);
    else
        YP_LAUNCH(0, k_front_heads_v3<0>, grid, block, (size_t)ts * 6 * sizeof(float), st, ha, cb,
                  params->conf_thr, t_lo, t_hi);
    // (network input size = grid x stride of the first head)
    return finish_v3(w, params, n_sets, (int)spi, ha.tpi, ts, d_out_rows, out_cap, d_out_count,
                     d_stats, st, (float)heads[0].grid_w * heads[0].stride_x,
                     (float)heads[0].grid_h * heads[0].stride_y);
}

int yp_detect_yolov3(const YpHead *heads, int n_heads, const YpParams *params, float *d_out_rows,
                     int64_t out_cap, int32_t *d_out_count, int32_t *d_stats, void *d_workspace,
                     size_t workspace_bytes, void *stream)
{
    return detect_v3_sets(heads, n_heads, 1, params, d_out_rows, out_cap, d_out_count, d_stats,
                          d_workspace, workspace_bytes, stream);
}

int yp_detect_yolov3_multi(const YpHead *heads, int n_heads, int n_sets, const YpParams *params,
                           float *d_out_rows, int64_t out_cap, int32_t *d_out_count,
                           int32_t *d_stats, void *d_workspace, size_t workspace_bytes,
                           void *stream)
{
    return detect_v3_sets(heads, n_heads, n_sets, params, d_out_rows, out_cap, d_out_count,
                          d_stats, d_workspace, workspace_bytes, stream);
}

// ---- the entry point that allocates nothing -------------------------------------------
// Scratch of yp_detect_yolov3_noalloc: the workspace yp_detect_yolov3 takes from its caller,
// held by the library instead.  Every GPU the library runs on has its own copy of this array
// at its own address; host code never uses the array's name as an address, only the pointer
// hipGetSymbolAddress returns for the device that was current when it was asked.
static __device__ __align__(256) unsigned char scratch[YP_NOALLOC_SCRATCH_BYTES];

namespace {
constexpr int kMaxDevices = 64;
void *g_scratch_ptr[kMaxDevices];        // per device, filled once by noalloc_scratch()
std::mutex g_scratch_mutex;

// Device address of `scratch` on the current device (looked up once per device).
int noalloc_scratch(void **out)
{
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= kMaxDevices) return YP_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void *ptr = nullptr;
        e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return (int)e;
        if (!ptr || (reinterpret_cast<uintptr_t>(ptr) % kAlign) != 0) return YP_ERR_WORKSPACE;
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return YP_OK;
}
}  // namespace

size_t yp_noalloc_scratch_bytes(void) { return (size_t)YP_NOALLOC_SCRATCH_BYTES; }

int yp_noalloc_prepare(void)
{
    void *ptr = nullptr;
    return noalloc_scratch(&ptr);
}

int yp_detect_yolov3_noalloc(const YpHead *heads, int n_heads, const YpParams *params,
                             float *d_out_rows, int64_t out_cap, int32_t *d_out_count,
                             int32_t *d_stats, void *stream)
{
    if (!params || params->batch <= 0 || params->num_classes <= 0) return YP_ERR_ARG;
    // the size test comes first: a problem beyond the static buffer is refused before anything
    // is launched (0 = the geometry itself is invalid or beyond the compiled limits)
    const size_t need = yp_detect_yolov3_workspace_bytes(heads, n_heads, params->batch,
                                                         params->num_classes);
    if (need == 0) return YP_ERR_ARG;
    if (need > (size_t)YP_NOALLOC_SCRATCH_BYTES) return YP_ERR_SCRATCH;
    void *ptr = nullptr;
    const int rc = noalloc_scratch(&ptr);
    if (rc != YP_OK) return rc;
    // every array of the workspace is written by the kernels before it is read (tile counts by
    // the front kernel, segment records and flags by the NMS kernel), so nothing has to be
    // cleared between calls; what an earlier, larger call left behind is never looked at
    return detect_v3_sets(heads, n_heads, 1, params, d_out_rows, out_cap, d_out_count, d_stats, ptr,
                          (size_t)YP_NOALLOC_SCRATCH_BYTES, stream);
}

size_t yp_rows_workspace_bytes(int batch, int rows_per_image, int segments_per_image)
{
    if (batch <= 0 || rows_per_image <= 0 || segments_per_image <= 0) return 0;
    const long long tpi = ((long long)rows_per_image + kRowTile - 1) / kRowTile;
    if (!slots_fit(batch, tpi * kRowTile)) return 0;
    return carve(nullptr, (size_t)batch * tpi * kRowTile, (size_t)batch * tpi,
                 (size_t)batch * segments_per_image, nullptr);
}

// Workspace of the reference-mode entries: the arrays of yp_rows_workspace_bytes(batch, rows, 1)
// without those of the graph path, which only YP_MODE_YOLOV3 uses.
static size_t ref_workspace_bytes(int batch, int rows_per_image)
{
    if (batch <= 0 || rows_per_image <= 0) return 0;
    const long long tpi = ((long long)rows_per_image + kRowTile - 1) / kRowTile;
    if (!slots_fit(batch, tpi * kRowTile)) return 0;
    return carve(nullptr, (size_t)batch * tpi * kRowTile, (size_t)batch * tpi, (size_t)batch, nullptr, false);
}

size_t yp_detect_reference_workspace_bytes(int batch, int grid_h, int grid_w, int num_anchors)
{
    if (grid_h <= 0 || grid_w <= 0 || num_anchors <= 0) return 0;
    const long long n = (long long)grid_h * grid_w * num_anchors;
    if (n >= (1ll << 28)) return 0;
    return ref_workspace_bytes(batch, (int)n);
}

int yp_write_results(const float *d_rows, int rows_per_image, const YpParams *params,
                     float *d_out_rows, int64_t out_cap, int32_t *d_out_count, int32_t *d_stats,
                     void *d_workspace, size_t workspace_bytes, void *stream)
{
    if (!params_ok(params, YP_MODE_YOLOV3) || !d_rows || !d_out_rows || !d_out_count ||
        out_cap < 0 || rows_per_image <= 0)
        return YP_ERR_ARG;
    if (params->flags & YP_FLAG_RESCALE) return YP_ERR_ARG;   // decoded rows carry no input size
    if ((reinterpret_cast<uintptr_t>(d_out_rows) & 15) != 0) return YP_ERR_ARG;
    const int tpi = (rows_per_image + kRowTile - 1) / kRowTile;
    const long long spi = (long long)tpi * kRowTile;
    if (!slots_fit(params->batch, spi)) return YP_ERR_UNSUPPORTED;
    const size_t slots = (size_t)params->batch * spi, tiles = (size_t)params->batch * tpi;
    Workspace w;
    const size_t need = yp_rows_workspace_bytes(params->batch, rows_per_image, params->num_classes);
    if (!workspace_ok(d_workspace, workspace_bytes, need)) return YP_ERR_WORKSPACE;
    carve(d_workspace, slots, tiles, (size_t)params->batch * params->num_classes, &w);
    CandBuf cb = {w.cand_box, w.cand_key, w.cand_aux, w.cand_cls, w.tile_count, tpi, kRowTile, w.ctrl};
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = yp_tile_lds_bytes(5 + params->num_classes);
    if (lds)
        YP_LAUNCH(0, k_front_rows_v3<true>, dim3(tpi, params->batch), dim3(kRowTile), lds, st, d_rows,
                  rows_per_image, params->num_classes, cb, params->conf_thr);
    else
        YP_LAUNCH(0, k_front_rows_v3<false>, dim3(tpi, params->batch), dim3(kRowTile), 0, st, d_rows,
                  rows_per_image, params->num_classes, cb, params->conf_thr);
    return finish_v3(w, params, 1, (int)spi, tpi, kRowTile, d_out_rows, out_cap, d_out_count,
                     d_stats, st);
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
    const size_t need = ref_workspace_bytes(params->batch, (int)n);
    if (!workspace_ok(d_workspace, workspace_bytes, need)) return YP_ERR_WORKSPACE;
    Workspace w;
    carve(d_workspace, (size_t)params->batch * spi, (size_t)params->batch * tpi,
          (size_t)params->batch, &w, false);
    CandBuf cb = {w.cand_box, w.cand_key, w.cand_aux, w.cand_cls, w.tile_count, tpi, kRowTile, w.ctrl};
    AnchorTable at = {};
    for (int i = 0; i < 2 * head->num_anchors; ++i) at.anc[i] = head->anchors[i];
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = yp_tile_lds_bytes(5 + params->num_classes);
    if (lds && params->num_classes == 5)
        YP_LAUNCH(0, (k_front_grid_ref<true, 5>), dim3(tpi, params->batch), dim3(kRowTile), lds, st,
                  head->data, head->grid_h, head->grid_w, head->num_anchors, params->num_classes, at,
                  cb, params->conf_thr, params->im_h, params->im_w);
    else if (lds)
        YP_LAUNCH(0, (k_front_grid_ref<true, 0>), dim3(tpi, params->batch), dim3(kRowTile), lds, st,
                  head->data, head->grid_h, head->grid_w, head->num_anchors, params->num_classes, at,
                  cb, params->conf_thr, params->im_h, params->im_w);
    else
        YP_LAUNCH(0, (k_front_grid_ref<false, 0>), dim3(tpi, params->batch), dim3(kRowTile), 0, st,
                  head->data, head->grid_h, head->grid_w, head->num_anchors, params->num_classes, at,
                  cb, params->conf_thr, params->im_h, params->im_w);
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
    const size_t need = ref_workspace_bytes(params->batch, rows_per_image);
    if (!workspace_ok(d_workspace, workspace_bytes, need)) return YP_ERR_WORKSPACE;
    Workspace w;
    carve(d_workspace, (size_t)params->batch * spi, (size_t)params->batch * tpi,
          (size_t)params->batch, &w, false);
    CandBuf cb = {w.cand_box, w.cand_key, w.cand_aux, w.cand_cls, w.tile_count, tpi, kRowTile, w.ctrl};
    hipStream_t st = (hipStream_t)stream;
    YP_LAUNCH(0, k_front_split_ref, dim3(tpi, params->batch), dim3(kRowTile), 0, st, d_center, d_wh,
              d_obj, d_cls, rows_per_image, params->num_classes, cb, params->conf_thr, params->im_h,
              params->im_w);
    return finish_ref(w, params, (int)spi, tpi, kRowTile, d_out_boxes, d_out_scores, d_out_classes,
                      d_out_index, d_out_counts, d_surv_counts, st);
}

#ifdef YP_STAMPS
/* diagnostic build only: device buffer of 32 u64 stamps per NMS segment (or null) */
int yp_debug_set_stamps(void *d_buf)
{
    g_dbg = static_cast<unsigned long long *>(d_buf);
    return YP_OK;
}
/* 4 u64 stamps per workgroup of the heads front kernel (start, loaded, ranked, end) */
int yp_debug_set_front_stamps(void *d_buf)
{
    g_dbg_front = static_cast<unsigned long long *>(d_buf);
    return YP_OK;
}
#endif

int yp_profile_start(int max_calls)
{
    if (max_calls <= 0 || max_calls > (1 << 16) || g_prof.on) return YP_ERR_ARG;
    g_prof.ev = new hipEvent_t[kProfEvents * (size_t)max_calls];
    g_prof.launched = new unsigned char[(size_t)max_calls]();
    for (int i = 0; i < kProfEvents * max_calls; ++i) {
        const hipError_t e = hipEventCreate(&g_prof.ev[i]);
        if (e != hipSuccess) return (int)e;
    }
    g_prof.max_calls = max_calls;
    g_prof.calls = 0;
    g_prof.on = true;
    return YP_OK;
}

int yp_profile_stop_ex(double *ms, int *calls)
{
    if (!g_prof.on || !ms || !calls) return YP_ERR_ARG;
    g_prof.on = false;
    double acc[kProfKernels] = {};
    int rc = YP_OK;
    for (int c = 0; c < g_prof.calls && rc == YP_OK; ++c)
        for (int k = 0; k < kProfKernels; ++k) {
            if (!(g_prof.launched[c] & (1u << k))) continue;
            float t = 0.f;
            hipError_t e = hipEventSynchronize(g_prof.ev[kProfEvents * c + 2 * k + 1]);
            if (e == hipSuccess)
                e = hipEventElapsedTime(&t, g_prof.ev[kProfEvents * c + 2 * k],
                                        g_prof.ev[kProfEvents * c + 2 * k + 1]);
            if (e != hipSuccess) { rc = (int)e; break; }
            acc[k] += t;
        }
    *calls = g_prof.calls;
    // averages per CALL (a kernel a call did not launch counts as zero time)
    acc[1] += acc[5];
    acc[3] += acc[6];
    for (int k = 0; k < kProfReported; ++k) ms[k] = g_prof.calls ? acc[k] / g_prof.calls : 0.0;
    for (int i = 0; i < kProfEvents * g_prof.max_calls; ++i) (void)hipEventDestroy(g_prof.ev[i]);
    delete[] g_prof.ev;
    delete[] g_prof.launched;
    g_prof.ev = nullptr;
    g_prof.launched = nullptr;
    g_prof.max_calls = g_prof.calls = 0;
    return rc;
}

int yp_profile_stop(double *front_ms, double *nms_ms, double *gather_ms, int *calls)
{
    if (!front_ms || !nms_ms || !gather_ms) return YP_ERR_ARG;
    double ms[kProfReported];
    const int rc = yp_profile_stop_ex(ms, calls);
    if (rc != YP_OK) return rc;
    *front_ms = ms[0];
    *nms_ms = ms[1] + ms[3] + ms[4];
    *gather_ms = ms[2];
    return YP_OK;
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
