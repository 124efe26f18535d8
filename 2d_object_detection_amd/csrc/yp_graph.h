// yp_graph.h -- greedy NMS of LARGE segments as a dependency graph, spread over many
// workgroups.  Included by yolopost.hip inside its anonymous namespace (it uses that file's
// wave helpers, suppress_words and SegArgs).
//
// Greedy NMS over rows sorted by score keeps row i iff no KEPT row j < i suppresses it.  That
// is a recurrence over the graph "j suppresses i, j ranked before i", and it can be evaluated
// without walking the rows in order:
//
//   k_segment_nms   (one workgroup per segment, the existing kernel) gathers and sorts the
//                   segment as before; for a large segment it then only builds a spatial index
//                   -- the boxes binned by (size class, cell row, cell column) of their centre
//                   with a counting sort -- and leaves.
//   k_graph_pairs   many workgroups per segment.  A workgroup (or one of its waves) takes 32 boxes
//                   that are neighbours in the index ("columns", each held by two lanes), looks up
//                   the bins of their own and of larger size classes whose boxes can overlap any of
//                   them, stages those "rows" in LDS and tests every row against the columns with
//                   the division-free test of the other paths, two rows per packed instruction.
//                   For every box it records (atomics) how many earlier-ranked boxes suppress it,
//                   the earliest of them and up to seven more.
//   k_graph_resolve one workgroup per segment iterates  "a suppressor is KEPT -> SUPPRESSED;
//                   all suppressors SUPPRESSED -> KEPT"  over the undecided columns until none is
//                   left (the lowest-ranked undecided column can always be decided, so at most
//                   2K rounds; a handful in practice), re-deriving a column's suppressor list
//                   from the index when the recorded ones are used up, and emits the kept rows in
//                   rank order.
//
// Nothing is skipped that could change a verdict.  A row r and a column c can only reach the
// threshold if they overlap by a good part of the row: the union is at least the row's area and
// the intersection is at most (overlap along x) * (the row's height), so IoU <= overlap_x / w_r and
// likewise along y -- IoU >= thr needs an overlap of thr * w_r along x and thr * h_r along y (widths
// and heights as the IoU counts them, the +1 included).  The column's box then reaches into the
// row's CORE, the row's box with 0.9 * thr of its width and height (at most half) cut off on every
// side; a pair whose column misses the core has IoU < 0.9 * thr, far outside any rounding of the
// quotient.  So a row is dropped when its core misses the bounding box of the batch's columns, and
// a bin is left out when no box of its class whose centre lies in it can have a core that reaches
// that bounding box (with 2 px and the coordinates' rounding as slack), or when it belongs to a
// size class whose boxes cannot reach the threshold against the batch's classes (IoU <=
// min(w_i/w_j, h_i/h_j) for w_i <= w_j ...).  All of this needs a positive threshold inside the
// guard-band range and boxes with ordered corners; segments that do not qualify stay on the
// chunked path of k_segment_nms.
#pragma once

constexpr int kGraphMin = kGraphMinCand;  // segments with more candidates take this path
constexpr int kGraphBins = 4096;          // bins of a segment's index (all classes together)
constexpr int kGraphClasses = 8;          // size classes (powers of two of max(width, height))
constexpr int kGraphEdges = 8;            // recorded suppressors per column (the earliest + 7)
// Columns of a k_graph_pairs batch: with 32, every column is held by two lanes that test it
// against different rows at the same time.  A narrower batch covers a smaller patch of the plane,
// so fewer rows reach it; the number of batches, and with it the work of setting one up, grows
// instead.  (Measured on an MI355X, pair kernel per 32 images at 1920x1280, batches of 16 / 32 /
// 64 columns: 29 500 candidates per segment 2.98 / 2.55 / 2.24 ms, 2 000 per segment 95 / 95 /
// 125 us, clustered scenes of 4 700 per segment 309 / 321 / 401 us.)
// Candidate rows a wave fetches per trip to memory, and the waves per SIMD k_graph_pairs is
// compiled for (128 vector registers, no scratch).  A wave has two tiles in registers: the one it
// sifts and tests, and the next one, requested before that work starts.  (Measured on an MI355X
// before the tiles were fetched ahead, pair kernel per 32 images at 1920x1280: tiles of 256 rows
// at 3 waves per SIMD 4.47 ms / 368 us / 1.10 ms (29 500 candidates per segment / 2 000 /
// clustered scenes), 128 rows at 4 waves 4.57 ms / 324 us / 0.95 ms, 64 rows at 4 waves 4.80 ms /
// 309 us / 0.89 ms; with the next tile in flight 64-row tiles fit the register budget and gave
// 4.21 ms / 185 us / 0.77 ms.)
constexpr int kGraphRowTile = 64;
constexpr int kGraphPairWaves = 4;
constexpr int kGraphWaveModeMax = 8192;   // segments up to this size: one batch per wave of k_graph_pairs
constexpr int kGraphListCache = 512;      // graph segments k_graph_pairs looks at per pass over the list
constexpr int kGraphThreads = 256;        // k_graph_pairs workgroup
constexpr int kGraphWaves = kGraphThreads / 64;
constexpr int kGraphRanges = 64;          // bin ranges a wave looks up per pass

enum { kUndecided = 0, kKept = 1, kSuppressed = 2 };

// Index of one large segment (written by k_segment_nms, read by the two kernels below).
struct GraphSeg {
    int K;          // candidates; 0 = the segment did not take this path
    int table;      // its bin table in the pool: kGraphBins + 1 offsets
    int e_min;      // biased exponent of the smallest box size of the segment
    int d;          // classes further apart than this cannot reach the threshold
    float x0, y0;   // smallest centre coordinates
    float margin;   // slack added to every reach (2 px + rounding of the coordinates)
    float shrink;   // a row is looked at through its core: the box without this fraction of its width
                    // and height on every side (see graph_walk)
    int n_bins;
    float invx[kGraphClasses];     // 1 / cell width and 1 / cell height of the class's grid
    float invy[kGraphClasses];
    float half[kGraphClasses];     // upper bound of half the size of a box's CORE in the class
    int nx[kGraphClasses], ny[kGraphClasses], base[kGraphClasses];
    // Batches of k_graph_pairs never hold boxes of two size classes (the last boxes of one class's
    // grid and the first of the next lie at opposite ends of the plane, and a batch of both would
    // have to look at nearly every row): class c owns the index positions [cpos[c], cpos[c + 1])
    // and the batches [cbat[c], cbat[c + 1]), the last of which may be partly filled.
    int n_batches, cols;        // (cols: columns per batch, graph_cols(K))
    int cpos[kGraphClasses + 1], cbat[kGraphClasses + 1];
};

struct GraphArgs {
    int per_class, C, spi, max_out;
    float thr;
    const GraphSeg *gseg;        // [n_seg]
    const int *bin_tables;       // pool of (kGraphBins + 1)-entry offset tables
    const float4 *xraw;          // [B*spi] prepared corners in index order (reuses the key area)
    const int *xidx;             // [B*spi] rank of the box at an index position
    unsigned char *status;       // [B*spi] per rank: kUndecided / kKept / kSuppressed
    int *gcnt;                   // [B*spi] per index position: suppressors found
    int *gedge;                  // [B*spi * kGraphEdges] per index position: their ranks
    const int *sslot;            // [B*spi] rank -> candidate slot
    const int *seg_region;       // [n_seg]
    const int *glist;            // graph segments of the launch, by the order their tables were taken
    int *kept_slot;              // [B*spi]
    int *seg_kept;               // [n_seg]
    unsigned int *seg_pairs;     // [n_seg]
    unsigned int *seg_ties;      // [n_seg]
    int *work;                   // [4 * B*spi] two work lists per segment (the sorted-box area, which
                                 // nothing reads after the index was built)
    unsigned long long *dbg;     // diagnostic builds only (YP_STAMPS): 16 counters of k_graph_pairs
};

// Diagnostic build: wave-clock ticks per phase of a walk, summed over the waves of the launch.
#ifdef YP_STAMPS
#define GW_T(k)                                                \
    do {                                                       \
        const unsigned int tn_ = (unsigned int)__builtin_readcyclecounter(); \
        if ((threadIdx.x & 63) == 0) atomicAdd(&ws.dbg[k], tn_ - gw_prev); \
        gw_prev = tn_;                                         \
    } while (0)
#define GW_N(k, n) do { if ((threadIdx.x & 63) == 0) atomicAdd(&ws.dbg[k], (unsigned int)(n)); } while (0)
#else
#define GW_T(k) do { } while (0)
#define GW_N(k, n) do { } while (0)
#endif

// Size of a box for the class test: the larger of the width and height terms the IoU uses.
template <int MODE>
__device__ __forceinline__ float graph_box_size(const float4 &c)
{
    const float plus1 = MODE == YP_MODE_YOLOV3 ? 1.0f : 0.0f;
    const float w = c.z - c.x + plus1, h = c.w - c.y + plus1;
    return w > h ? w : h;
}

__device__ __forceinline__ int graph_class(float size, int e_min)
{
    const int e = (int)((__float_as_uint(size) >> 23) & 255u) - e_min;
    return e < 0 ? 0 : (e >= kGraphClasses ? kGraphClasses - 1 : e);
}

// Cell index of a coordinate along one axis of a class grid: monotone in v (subtraction of a
// constant, multiplication by a non-negative constant, floor, clamp), which is what makes "every
// box whose centre lies in [a, b] sits in a cell of [cell(a), cell(b)]" hold without rounding
// caveats.  Cells are laid out row by row, every other row backwards, so that consecutive
// index positions stay neighbours in the plane across the end of a row.
__device__ __forceinline__ int graph_cell(float v, float origin, float inv, int n)
{
    float t = (v - origin) * inv;
    t = t > 0.0f ? t : 0.0f;                  // (also a NaN)
    const float top = (float)(n - 1);
    t = t < top ? t : top;
    return (int)t;
}

__device__ __forceinline__ float wave_min_f(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = vmin(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float wave_max_f(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = vmax(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ int wave_min_i(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}
__device__ __forceinline__ int wave_max_i(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

constexpr int kGraphBuf = kGraphRowTile + 64;   // useful rows a wave collects before testing
// Columns of a k_graph_pairs batch: 32 (each held by two lanes that take different rows of a step)
// for segments of fewer than kGraphWideMin candidates, 64 for larger ones.
constexpr int kGraphColsSmall = 32;
constexpr int kGraphWideMin = 16384;
static_assert(kGraphBuf % (4 * (64 / kGraphColsSmall)) == 0, "padding of a flush stays inside the buffer");
__device__ __forceinline__ int graph_cols(int K) { return K >= kGraphWideMin ? 64 : kGraphColsSmall; }

// Per-wave LDS of a walk.
struct GraphWaveShared {
    float4 row[kGraphBuf];          // useful rows: prepared corners
    float area[kGraphBuf];          // their area terms
    int rank[kGraphBuf];            // their ranks
    int pc[kGraphBuf];              // index position | size class << 24
    int rbeg[kGraphRanges];         // bin ranges of the current pass: first index position
    int rpre[kGraphRanges + 1];     // exclusive prefix of their lengths
#ifdef YP_STAMPS
    unsigned int dbg[12];           // diagnostic build: ticks per phase
#endif
};

// Records "the row ranked `suppressor` suppresses the box at index position `target`": the
// count, the earliest suppressor and the first kGraphEdges-1 that arrive (any order).
__device__ __forceinline__ void graph_record(int *gcnt, int *gedge, int target, int suppressor)
{
    const int k = atomicAdd(&gcnt[target], 1);
    if (k < kGraphEdges - 1) gedge[(size_t)target * kGraphEdges + 1 + k] = suppressor;
    atomicMin(&gedge[(size_t)target * kGraphEdges], suppressor);
}

// One batch: the 64 columns held by the lanes (corners `cj`, rank `my_rank`, index position
// `my_pos`, valid where `live`) against the rows of the index that can overlap any of them.
//
// REBUILD == false (k_graph_pairs): every unordered pair of the segment has to be tested once,
// so the batch looks at its own size class and the LARGER ones within reach -- small boxes are
// many and dense, and 64 neighbours of them reach few larger boxes, whereas 64 large boxes
// between them would reach every small one.  A suppressing pair is recorded at its later-ranked
// box with graph_record: at the column when the row is ranked earlier, at the row when the
// column is and the row's class is larger (within one class the pair comes up a second time with
// the roles swapped).  The walk is shared by the `n_split` waves of a workgroup: wave
// `split` takes every n_split-th tile of candidate rows.
//
// REBUILD == true (k_graph_resolve): the columns need ALL their suppressors that are not
// suppressed themselves: smaller and larger classes, rows with status kSuppressed left out.
// Per lane the count is returned, the ranks go to `edge_out` (the earliest in [0], then up to
// kGraphEdges-1 in the order found), and kept_hit tells that one of them is already kept.
// (k_graph_pairs passes kGraphEdges words of LDS per lane as `edge_out` too: the suppressors of the
// lane's own column collect there during the walk.)
//
// Candidate rows are looked at by one lane each when they are fetched: a row that cannot touch
// the batch's bounding box, whose class is out of order for the pair rule, or that is already
// suppressed (REBUILD) is dropped there; the others are packed into LDS, and whenever enough
// have gathered they are tested against the columns, four per lane and step (two packed tests).
// With COLS < 64 columns per batch (k_graph_pairs only) lane l holds column l % COLS and takes
// the rows l / COLS, 64 / COLS + l / COLS, ... of every group of 4 * 64 / COLS rows.
template <int MODE, bool REBUILD, int COLS>
__device__ __forceinline__ int graph_walk(const GraphSeg &gs, const int *off, const float4 *xraw,
                                          const int *xidx, const unsigned char *status, int *gcnt,
                                          int *gedge, const float4 &cj, int my_rank, int my_pos,
                                          bool live, const YpThr &thr, GraphWaveShared &ws, int split,
                                          int n_split, int *edge_out, unsigned int &pairs,
                                          unsigned int &ties, bool &kept_hit)
{
#ifdef YP_STAMPS
    unsigned int gw_prev = (unsigned int)__builtin_readcyclecounter();
    GW_N(8, 1);
#endif
    const int lane = threadIdx.x & 63;
    constexpr int REP = 64 / COLS;
    const int rslot = REP == 1 ? 0 : lane / COLS;     // which of the REP rows of a step this lane takes
    const YpBox bj = {cj.x, cj.y, cj.z, cj.w};
    const YpPrep pj = yp_prep<MODE>(bj);
    const float4 rj = make_float4(pj.lo0, pj.lo1, pj.hi0, pj.hi1);
    const u64 live_mask = __ballot(live);
    // classes and extent of the batch (uniform)
    const int my_class = graph_class(graph_box_size<MODE>(cj), gs.e_min);
    const int cA = wave_min_i(live ? my_class : kGraphClasses), cB = wave_max_i(live ? my_class : -1);
    const float inf = __builtin_inff();
    const float bxlo = wave_min_f(live ? pj.lo0 : inf), bxhi = wave_max_f(live ? pj.hi0 : -inf);
    const float bylo = wave_min_f(live ? pj.lo1 : inf), byhi = wave_max_f(live ? pj.hi1 : -inf);
    const int n_live = __popcll(__ballot(live) & (COLS == 64 ? ~0ull : (1ull << (COLS & 63)) - 1ull));
    int cnt = 0, first = 0x7fffffff;
    kept_hit = false;
    if (cB < 0) return 0;
    const int c_first = REBUILD ? cA - gs.d : cA;
    const int c_lo = c_first > 0 ? c_first : 0;
    const int c_hi = cB + gs.d < kGraphClasses - 1 ? cB + gs.d : kGraphClasses - 1;

    int n_buf = 0;      // useful rows waiting in LDS (uniform)
    // test the waiting rows against the columns, four rows per lane and step
    auto flush = [&]() {
        GW_T(4);
        GW_N(10, (n_buf + 4 * REP - 1) / (4 * REP));
        // pad to a multiple of four (times REP) with rows that cannot record anything
        constexpr int G = 4 * REP;
        if (lane < G) {
            const int q = n_buf + lane;
            if (q < ((n_buf + G - 1) & ~(G - 1))) {
                ws.row[q] = make_float4(3e38f, 3e38f, 3e38f, 3e38f);
                ws.area[q] = 1.0f;
                ws.rank[q] = 0x7fffffff;
                ws.pc[q] = 0;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int r0 = 0; r0 < n_buf; r0 += G) {
            float4 ci[4];
            float ai[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = r0 + u * REP + rslot;
                ci[u] = ws.row[q];
                ai[u] = ws.area[q];
            }
            // (packed tests, two rows per instruction: the segment qualified with x2 >= x1, y2 >= y1
            // and finite areas, so every union is >= 1; the padding rows decide "no")
            u64 words[4];
            unsigned int nt = 0u;
            pair2_words<MODE>(rj, pj.area, ci[0], ci[1], ai[0], ai[1], thr, live_mask, live_mask, words[0],
                              words[1], nt);
            pair2_words<MODE>(rj, pj.area, ci[2], ci[3], ai[2], ai[3], thr, live_mask, live_mask, words[2],
                              words[3], nt);
            ties += lane == 0 ? nt : 0u;
            if ((words[0] | words[1] | words[2] | words[3]) == 0ull) continue;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (words[u] == 0ull) continue;
                const bool sup = (words[u] >> lane) & 1ull;
                // (rank and position of the row are only looked up for the few rows that suppress)
                const int q = r0 + u * REP + rslot;
                const int rk = ws.rank[q];
                if (REBUILD) {
                    // only rows ranked before the column count (this also drops the column itself)
                    if (sup && rk < my_rank) {
                        if (cnt < kGraphEdges - 1) edge_out[1 + cnt] = rk;
                        first = rk < first ? rk : first;
                        ++cnt;
                        if (status[rk] == kKept) kept_hit = true;
                    }
                } else if (sup) {
                    const int pc = ws.pc[q];
                    const int rc = pc >> 24;
                    if (rk < my_rank) {
                        // (suppressors of the lane's own column wait in LDS and are recorded together
                        // when the batch is done: one trip to memory per column instead of one per pair)
                        if (rc >= my_class) {
                            if (cnt < kGraphEdges - 1) edge_out[1 + cnt] = rk;
                            first = rk < first ? rk : first;
                            ++cnt;
                        }
                    } else if (rk > my_rank && rc > my_class) {
                        graph_record(gcnt, gedge, pc & 0xffffff, my_rank);
                    }
                }
            }
        }
        if (lane == 0) pairs += (unsigned int)(n_buf * n_live);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();   // the buffer is refilled
        n_buf = 0;
        GW_T(5);
    };

    // ranges = (class, cell row) pairs in reach, each a run of consecutive cells = one contiguous
    // run of index positions; numbered class-major and handed to the lanes 64 at a time.  Lane
    // c < kGraphClasses works out class c's window once; the lanes then pick theirs from it.
    int w_n = 0, w_iy0 = 0, w_ix0 = 0, w_ix1 = 0, w_nx = 1, w_base = 0;
    if (lane >= c_lo && lane <= c_hi && gs.nx[lane] > 0) {
        const float reach = gs.half[lane] + gs.margin;
        w_nx = gs.nx[lane];
        w_base = gs.base[lane];
        w_iy0 = graph_cell(bylo - reach, gs.y0, gs.invy[lane], gs.ny[lane]);
        w_n = graph_cell(byhi + reach, gs.y0, gs.invy[lane], gs.ny[lane]) - w_iy0 + 1;
        w_ix0 = graph_cell(bxlo - reach, gs.x0, gs.invx[lane], w_nx);
        w_ix1 = graph_cell(bxhi + reach, gs.x0, gs.invx[lane], w_nx);
    }
    const int w_incl = wave_inclusive_scan(w_n, lane);
    const int n_ranges = __builtin_amdgcn_readlane(w_incl, kGraphClasses - 1);
    GW_T(0);
    int tile_base = 0;   // candidate tiles of the batch in earlier passes (tiles are numbered over all passes)
    for (int g0 = 0; g0 < n_ranges; g0 += kGraphRanges) {
        // lane l: range g0 + l
        int beg = 0, len = 0;
        {
            const int want = g0 + lane;
            int iy = 0, ix0 = 0, ix1 = 0, nx = 1, base = 0;
            bool found = false;
#pragma unroll
            for (int c = 0; c < kGraphClasses; ++c) {
                const int hi = __builtin_amdgcn_readlane(w_incl, c), n = __builtin_amdgcn_readlane(w_n, c);
                if (want >= hi - n && want < hi) {
                    found = true;
                    iy = __builtin_amdgcn_readlane(w_iy0, c) + (want - (hi - n));
                    ix0 = __builtin_amdgcn_readlane(w_ix0, c);
                    ix1 = __builtin_amdgcn_readlane(w_ix1, c);
                    nx = __builtin_amdgcn_readlane(w_nx, c);
                    base = __builtin_amdgcn_readlane(w_base, c);
                }
            }
            if (found) {
                const int b0 = base + iy * nx;
                const int ca = (iy & 1) ? nx - 1 - ix1 : ix0;     // (odd rows run backwards)
                const int cb = (iy & 1) ? nx - 1 - ix0 : ix1;
                beg = off[b0 + ca];
                len = off[b0 + cb + 1] - beg;
            }
        }
        const int incl = wave_inclusive_scan(len, lane);
        ws.rbeg[lane] = beg;
        ws.rpre[lane + 1] = incl;
        if (lane == 0) ws.rpre[0] = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int n_rows = __builtin_amdgcn_readfirstlane(__shfl(incl, 63, 64));
        GW_T(1);
        // The tiles of this pass that are this wave's: tile_base + i with (tile_base + i) % n_split
        // == split.  The loads of a tile are issued before the tile in front of it is sifted and
        // tested, so a trip to memory is hidden behind the pair tests instead of waited for.
        constexpr int T = kGraphRowTile / 64;
        const int n_tiles = (n_rows + kGraphRowTile - 1) / kGraphRowTile;
        // element e of the concatenated ranges -> its range by bisection of the prefix
        auto locate = [&](int t0, int (&src)[T]) {
            const int fill = n_rows - t0 < kGraphRowTile ? n_rows - t0 : kGraphRowTile;
#pragma unroll
            for (int k = 0; k < T; ++k) {
                const int e = t0 + k * 64 + lane;
                int lo = 0, hi = kGraphRanges - 1;     // largest l with rpre[l] <= e
#pragma unroll
                for (int st = 0; st < 6; ++st) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (ws.rpre[mid] <= e) lo = mid; else hi = mid - 1;
                }
                src[k] = k * 64 + lane < fill ? ws.rbeg[lo] + (e - ws.rpre[lo]) : 0;
            }
        };
        int i_tile = ((split - tile_base) % n_split + n_split) % n_split;
        int src_n[T];
        float4 c4_n[T];
        int r4_n[T];
        if (i_tile < n_tiles) {
            locate(i_tile * kGraphRowTile, src_n);
#pragma unroll
            for (int k = 0; k < T; ++k) {
                c4_n[k] = xraw[src_n[k]];
                r4_n[k] = xidx[src_n[k]];
            }
        }
        GW_T(2);
        while (i_tile < n_tiles) {
#ifdef YP_STAMPS
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the tile's loads have landed
            GW_T(3);
            GW_N(9, 1);
#endif
            const int t0 = i_tile * kGraphRowTile;
            const int fill = n_rows - t0 < kGraphRowTile ? n_rows - t0 : kGraphRowTile;
            int src[T];
            float4 c4[T];
            int r4[T];
#pragma unroll
            for (int k = 0; k < T; ++k) {
                src[k] = src_n[k];
                c4[k] = c4_n[k];
                r4[k] = r4_n[k];
            }
            i_tile += n_split;
            if (i_tile < n_tiles) {
                locate(i_tile * kGraphRowTile, src_n);
#pragma unroll
                for (int k = 0; k < T; ++k) {
                    c4_n[k] = xraw[src_n[k]];
                    r4_n[k] = xidx[src_n[k]];
                }
            }
            GW_T(2);
            // sift and pack
#pragma unroll
            for (int k = 0; k < T; ++k) {
                const int rc = graph_class(graph_box_size<MODE>(c4[k]), gs.e_min);
                // the row's core against the bounding box of the batch (see the note at the top)
                const float sx = gs.shrink * (c4[k].z - c4[k].x + 1.0f), sy = gs.shrink * (c4[k].w - c4[k].y + 1.0f);
                bool useful = k * 64 + lane < fill && !(c4[k].z + 1.0f - sx + gs.margin < bxlo) &&
                              !(c4[k].x + sx - gs.margin > bxhi + 1.0f) && !(c4[k].w + 1.0f - sy + gs.margin < bylo) &&
                              !(c4[k].y + sy - gs.margin > byhi + 1.0f);
                if (!REBUILD) useful = useful && rc >= cA;
                if (REBUILD && useful) useful = status[r4[k]] != kSuppressed;
                const u64 m = __ballot(useful);
                if (useful) {
                    const int q = n_buf + __popcll(m & lanes_below(lane));
                    const YpBox bi = {c4[k].x, c4[k].y, c4[k].z, c4[k].w};
                    ws.row[q] = c4[k];
                    ws.area[q] = yp_prep<MODE>(bi).area;
                    ws.rank[q] = r4[k];
                    ws.pc[q] = src[k] | (rc << 24);
                }
                n_buf += __popcll(m);
            }
            GW_T(4);
            if (n_buf > 64) flush();     // (room for the next tile: kGraphBuf - 64 >= a tile)
        }
        tile_base += n_tiles;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();       // the next pass overwrites the ranges
    }
    if (n_buf > 0) flush();
    if (REBUILD) {
        edge_out[0] = first;
    } else if (cnt > 0) {
        // the column's suppressors of this walk: room for all of them is taken with one atomic
        const int k0 = atomicAdd(&gcnt[my_pos], cnt);
        int *e = gedge + (size_t)my_pos * kGraphEdges;
        const int listed = cnt < kGraphEdges - 1 ? cnt : kGraphEdges - 1;
        for (int k = 0; k < listed; ++k)
            if (k0 + k < kGraphEdges - 1) e[1 + k0 + k] = edge_out[1 + k];
        atomicMin(&e[0], first);
    }
    GW_T(6);
    return cnt;
}

// Pair tests of every graph segment of the launch.  The unit of work is a "group": one batch of
// a segment whose row tiles the four waves of a workgroup share, or -- for segments of a few
// thousand candidates, where a batch is reached by two or three tiles and half the waves would
// idle -- four batches, one per wave.
template <int MODE, int COLS>
__global__ __launch_bounds__(kGraphThreads) __attribute__((amdgpu_waves_per_eu(kGraphPairWaves)))
void k_graph_pairs(GraphArgs a, const int *ctrl)
{
    static_assert(COLS == 64 || COLS == kGraphColsSmall, "the two batch widths");
    __shared__ GraphWaveShared s_ws[kGraphWaves];
    __shared__ GraphSeg s_gs;
    __shared__ int s_pedge[kGraphWaves][64][kGraphEdges];   // suppressors of a lane's column during a walk
    __shared__ int s_k[kGraphListCache];        // sizes of the segments of the current chunk of the list
    __shared__ int s_mine[kGraphListCache];     // those this workgroup has groups of: index << 16 | first group
    __shared__ int s_wsum[kGraphWaves];
    __shared__ int s_nmine;
    static_assert(kGraphListCache == 2 * kGraphThreads, "two segments per thread");
    static_assert(kGraphPairGrid < 65536, "a workgroup number fits 16 bits");
    const int n_graph = ctrl[1];                    // graph segments of this launch
    if (n_graph == 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const YpThr thr = yp_make_thr(a.thr);
    GraphWaveShared &ws = s_ws[wave];
#ifdef YP_STAMPS
    // (diagnostic build: wall-clock ticks per workgroup -- start, list built, end -- and its groups)
    unsigned long long *wg_dbg = a.dbg ? a.dbg + 32 + 4 * (size_t)blockIdx.x : nullptr;
    if (threadIdx.x == 0 && wg_dbg) wg_dbg[0] = wall_clock64();
    int dbg_groups = 0;
    if (lane < 12) ws.dbg[lane] = 0u;
#endif
    // One batch per wave where the launch has batches to spare (ctrl[2] counts its candidates in
    // units of 64).  (Measured per 32 images at 1920x1280: 2 000 candidates per segment 324 -> 232 us
    // with a batch per wave; 29 500 per segment 4.57 ms shared, 5.12 ms with a batch per wave.)
    const bool spare = ctrl[2] * (64 / kGraphColsSmall) >= 4 * (int)gridDim.x;
    // The groups of all segments are dealt to the workgroups round-robin as ONE sequence: segment
    // t's group g goes to workgroup (groups of the segments before t + g) % grid, so every
    // workgroup gets the same number of groups to within one.  A workgroup first works out, for 512
    // segments of the list at a time and two per thread (a prefix sum over their group counts),
    // which segments hold a group for it; its loop then only visits those -- two or three of the
    // 160 segments of a launch of 2 000-candidate segments, where looking at every segment in turn
    // cost more instructions than the pair tests.  (Handing the groups out from a counter instead,
    // to whichever workgroup is free, evened out the workgroups' run times but made every workgroup
    // work on the same segment at the same time: 3.97 -> 4.91 ms at 29 500 candidates per segment,
    // 212 -> 205 us at 2 000.)
    const unsigned int grid = gridDim.x;
    unsigned int dealt = 0u;                        // groups of the chunks before this one, modulo the grid
    for (int base = 0; base < n_graph; base += kGraphListCache) {
        const int n_chunk = n_graph - base < kGraphListCache ? n_graph - base : kGraphListCache;
        __syncthreads();                            // the lists of the chunk before are no longer read
        if (threadIdx.x == 0) s_nmine = 0;
        int ng[2], sum = 0;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = 2 * (int)threadIdx.x + u;
            // (a 32-bit byte offset from the uniform base: the address needs no 64-bit vector arithmetic)
            const unsigned int goff = (unsigned int)a.glist[i < n_chunk ? base + i : base] * (unsigned int)sizeof(GraphSeg);
            const char *gp = reinterpret_cast<const char *>(a.gseg) + goff;
            const int K = i < n_chunk ? *reinterpret_cast<const int *>(gp + offsetof(GraphSeg, K)) : 0;
            // (this instantiation takes the segments whose batches have COLS columns)
            const int n_batches = i < n_chunk && graph_cols(K) == COLS ? *reinterpret_cast<const int *>(gp + offsetof(GraphSeg, n_batches)) : 0;
            ng[u] = spare && K <= kGraphWaveModeMax ? (n_batches + kGraphWaves - 1) / kGraphWaves : n_batches;
            sum += ng[u];
            if (i < n_chunk) s_k[i] = K;
        }
        const int incl = wave_inclusive_scan(sum, lane);
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        int woff = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kGraphWaves; ++w) {
            woff += w < wave ? s_wsum[w] : 0;
            total += s_wsum[w];
        }
        unsigned int before = dealt + (unsigned int)(woff + incl - sum);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = 2 * (int)threadIdx.x + u;
            if (i < n_chunk) {
                const unsigned int first = (blockIdx.x + grid - before % grid) % grid;
                if (first < (unsigned int)ng[u]) s_mine[atomicAdd(&s_nmine, 1)] = (i << 16) | (int)first;
            }
            before += (unsigned int)ng[u];
        }
        dealt = (dealt + (unsigned int)total) % grid;
        __syncthreads();
        const int n_mine = s_nmine;
#ifdef YP_STAMPS
        if (threadIdx.x == 0 && wg_dbg && base == 0) wg_dbg[1] = wall_clock64();
        dbg_groups += n_mine;
#endif
        for (int m = 0; m < n_mine; ++m) {
            const int entry = __builtin_amdgcn_readfirstlane(s_mine[m]);
            const int t = entry >> 16, start = entry & 0xffff;
            const int K = __builtin_amdgcn_readfirstlane(s_k[t]);
            const bool wave_mode = spare && K <= kGraphWaveModeMax;
            const int per_wg = wave_mode ? kGraphWaves : 1;
            const int seg = a.glist[base + t];
            __syncthreads();                            // everyone is done with the previous s_gs
            {
                const int *src = reinterpret_cast<const int *>(a.gseg + seg);
                int *dst = reinterpret_cast<int *>(&s_gs);
                for (int i = threadIdx.x; i < (int)(sizeof(GraphSeg) / 4); i += kGraphThreads) dst[i] = src[i];
            }
            __syncthreads();
            const int n_batches = s_gs.n_batches;
            const int n_groups = (n_batches + per_wg - 1) / per_wg;
            // Where a group sits in its segment's index decides what it costs (the index is ordered by
            // size class, and small boxes look at all the larger ones), and with segments of equal
            // size the deal above gives a workgroup the same place in every segment it visits: the
            // groups of a segment are therefore taken in an order rotated by a per-segment amount.
            const int rot = (int)((((unsigned int)(base + t) * 2654435761u) >> 8) % (unsigned int)n_groups);
            const int b = a.per_class ? seg / a.C : seg;
            const size_t reg = (size_t)b * a.spi + a.seg_region[seg];
            const int *off = a.bin_tables + (size_t)s_gs.table * (kGraphBins + 1);
            const float4 *xraw = a.xraw + reg;
            const int *xidx = a.xidx + reg;
            unsigned int pairs = 0, ties = 0;
#ifdef YP_STAMPS
            const unsigned int gw_seg0 = (unsigned int)__builtin_readcyclecounter();
#endif
            for (int g0 = start; g0 < n_groups; g0 += gridDim.x) {
                const int grp = g0 + rot < n_groups ? g0 + rot : g0 + rot - n_groups;
                const int batch = wave_mode ? grp * kGraphWaves + wave : grp;
                if (batch >= n_batches) continue;       // (uniform over the wave; nothing below is a workgroup barrier)
                int bc = 0;                             // the size class that owns the batch
                while (bc + 1 < kGraphClasses && s_gs.cbat[bc + 1] <= batch) ++bc;
                const int p = s_gs.cpos[bc] + (batch - s_gs.cbat[bc]) * COLS + (lane & (COLS - 1));
                const bool live = p < s_gs.cpos[bc + 1];
                const float4 cj = xraw[live ? p : s_gs.cpos[bc]];
                const int my_rank = live ? xidx[p] : -1;
                bool kept_hit;
                graph_walk<MODE, false, COLS>(s_gs, off, xraw, xidx, nullptr, a.gcnt + reg, a.gedge + reg * kGraphEdges,
                                        cj, my_rank, p, live, thr, ws, wave_mode ? 0 : wave,
                                        wave_mode ? 1 : kGraphWaves, s_pedge[wave][lane], pairs, ties, kept_hit);
            }
#ifdef YP_STAMPS
            if (lane == 0) ws.dbg[7] += (unsigned int)__builtin_readcyclecounter() - gw_seg0;   // the segment's batch loop, everything
#endif
            if (lane == 0 && pairs) atomicAdd(&a.seg_pairs[seg], pairs);
            if (lane == 0 && ties) atomicAdd(&a.seg_ties[seg], ties);
        }
    }
#ifdef YP_STAMPS
    __builtin_amdgcn_wave_barrier();
    if (lane < 12 && a.dbg && ws.dbg[lane]) atomicAdd(a.dbg + lane, (unsigned long long)ws.dbg[lane]);
    __syncthreads();
    if (threadIdx.x == 0 && wg_dbg) {
        wg_dbg[2] = wall_clock64();
        wg_dbg[3] = (unsigned long long)dbg_groups;
    }
#endif
}

// Keep set of one graph segment from the recorded suppressors, then its kept rows in rank
// order.  One workgroup per graph segment of the launch.
template <int MODE>
__global__ __launch_bounds__(kSegThreads) void k_graph_resolve(GraphArgs a, const int *ctrl)
{
    __shared__ GraphWaveShared s_ws[kSegWaves];
    __shared__ int s_edge[kSegWaves][64][kGraphEdges];
    __shared__ GraphSeg s_gs;
    __shared__ int s_undecided, s_redo;
    // statuses of segments of up to kResolveLdsMax candidates, two bits each: a round reads the
    // status of every column and of up to eight suppressors of it, one scattered byte each from
    // memory otherwise (the global array is written all the same: the re-derivations read it)
    constexpr int kResolveLdsMax = 32768;
    __shared__ unsigned int s_st[kResolveLdsMax / 16];
    __shared__ int s_nwork;
    __shared__ u64 s_bal[kSegWaves];
    if ((int)blockIdx.x >= ctrl[1]) return;
    const int seg = a.glist[blockIdx.x];
    {
        const int *src = reinterpret_cast<const int *>(a.gseg + seg);
        int *dst = reinterpret_cast<int *>(&s_gs);
        for (int i = threadIdx.x; i < (int)(sizeof(GraphSeg) / 4); i += kSegThreads) dst[i] = src[i];
    }
    __syncthreads();
    const int K = s_gs.K;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool st_lds = K <= kResolveLdsMax;
    for (int i = tid; i < kResolveLdsMax / 16; i += kSegThreads) s_st[i] = 0u;   // (kUndecided, as in memory)
    static_assert(kUndecided == 0 && kKept == 1 && kSuppressed == 2, "two status bits, set once");
    const int b = a.per_class ? seg / a.C : seg;
    const size_t reg = (size_t)b * a.spi + a.seg_region[seg];
    const int *off = a.bin_tables + (size_t)s_gs.table * (kGraphBins + 1);
    const float4 *xraw = a.xraw + reg;
    const int *xidx = a.xidx + reg;
    unsigned char *status = a.status + reg;
    int *gcnt = a.gcnt + reg;
    int *gedge = a.gedge + reg * kGraphEdges;
    const YpThr thr = yp_make_thr(a.thr);
    unsigned int pairs = 0, ties = 0;
    const int n_batches = (K + 63) >> 6;
    auto st_get = [&](int r) -> int {
        return st_lds ? (int)((s_st[r >> 4] >> ((r & 15) * 2)) & 3u) : (int)status[r];
    };
    auto st_set = [&](int r, int v) {     // (a status goes from undecided to its final value once)
        status[r] = (unsigned char)v;
        if (st_lds) atomicOr(&s_st[r >> 4], (unsigned int)v << ((r & 15) * 2));
    };
    __syncthreads();
    // Rounds.  Statuses only ever go from undecided to their final value, so reading one that
    // another wave has just written is as good as reading it a round later.  Every round
    // decides at least the lowest-ranked undecided column (all its suppressors are decided), or
    // re-derives its list and decides it in the next: the loop ends after at most 2K rounds.
    // After its first round a round only visits the columns the round before left undecided (a
    // work list) instead of all of them: most columns are decided in the first two rounds.
    int *wl_a = a.work + reg * 4, *wl_b = wl_a + K;   // this round's / the next round's columns
#ifdef YP_STAMPS
    // (diagnostic build, thread 0: wall-clock ticks in the rounds' scans, the re-derivations and the
    // output, and their numbers, summed over the segments into a.dbg[16..])
    unsigned long long rs_t = wall_clock64(), rs_scan = 0, rs_redo = 0, rs_rounds = 0, rs_walks = 0;
#define RS_T(acc) do { const unsigned long long n_ = wall_clock64(); acc += n_ - rs_t; rs_t = n_; } while (0)
#else
#define RS_T(acc) do { } while (0)
#endif
#define YP_WL_PUSH(p) (wl_b[atomicAdd(&s_nwork, 1)] = (p))
    int n_work = K;     // columns this round visits: all of them, or the work list's
    for (int round = 0; round < 2 * K + 2; ++round) {
        if (tid == 0) {
            s_undecided = 0;
            s_redo = 0;
            s_nwork = 0;
        }
        __syncthreads();
        int undecided = 0, redo = 0;
        for (int q = tid; q < n_work; q += kSegThreads) {
            const int p = round > 0 ? wl_a[q] : q;
            const int r = xidx[p];
            if (st_get(r) != kUndecided) continue;
            const int cnt = gcnt[p];
            if (cnt == 0) { st_set(r, kKept); continue; }
            if (cnt < 0) {                                        // list used up: re-derive below
                ++redo; ++undecided;
                YP_WL_PUSH(p);
                continue;
            }
            const int4 *e4 = reinterpret_cast<const int4 *>(gedge + (size_t)p * kGraphEdges);
            const int4 ea = e4[0], eb = e4[1];
            const int e[kGraphEdges] = {ea.x, ea.y, ea.z, ea.w, eb.x, eb.y, eb.z, eb.w};
            const int listed = cnt < kGraphEdges - 1 ? cnt : kGraphEdges - 1;
            bool any_kept = false, all_sup = true;
#pragma unroll
            for (int k = 0; k < kGraphEdges; ++k) {
                if (k <= listed) {       // [0] the earliest, [1..listed] the recorded ones
                    const int st = st_get(e[k]);
                    any_kept = any_kept || st == kKept;
                    all_sup = all_sup && st == kSuppressed;
                }
            }
            if (any_kept) st_set(r, kSuppressed);
            else if (all_sup && cnt <= kGraphEdges - 1) st_set(r, kKept);
            else {
                ++undecided;
                if (all_sup) { gcnt[p] = -1; ++redo; }               // more suppressors than recorded
                YP_WL_PUSH(p);
            }
        }
        if (undecided) atomicAdd(&s_undecided, undecided);
        if (redo) atomicAdd(&s_redo, redo);
        __syncthreads();
#ifdef YP_STAMPS
        ++rs_rounds;
        if (tid == 0 && a.dbg && blockIdx.x == 0 && round < 10)   // (first segment: columns visited and ticks of the scan, per round)
            a.dbg[22 + round] = ((wall_clock64() - rs_t) << 32) | (unsigned long long)n_work;
#endif
        RS_T(rs_scan);
        if (s_undecided == 0) break;
        {   // (read before thread 0 clears the counter at the top of the loop)
            n_work = s_nwork;
            int *t = wl_a;
            wl_a = wl_b;
            wl_b = t;
        }
        if (s_redo != 0) {
            // re-derive the suppressor lists of the flagged columns from the index, leaving out
            // rows that are suppressed by now; a kept suppressor settles the column on the spot
            for (int batch = wave; batch < n_batches; batch += kSegWaves) {
                const int p = batch * 64 + lane;
                const bool in = p < K;
                const int my_rank = in ? xidx[p] : -1;
                const bool need = in && gcnt[p] < 0 && st_get(my_rank) == kUndecided;
                if (__ballot(need) == 0ull) continue;
#ifdef YP_STAMPS
                if (wave == 0) ++rs_walks;
#endif
                const float4 cj = xraw[in ? p : K - 1];
                bool kept_hit;
                int *le = s_edge[wave][lane];
                const int cnt = graph_walk<MODE, true, 64>(s_gs, off, xraw, xidx, status, nullptr, nullptr, cj,
                                                       my_rank, p, need, thr, s_ws[wave], 0, 1, le, pairs,
                                                       ties, kept_hit);
                if (need) {
                    if (kept_hit) st_set(my_rank, kSuppressed);
                    else if (cnt == 0) st_set(my_rank, kKept);
                    gcnt[p] = cnt;
                    int4 *e = reinterpret_cast<int4 *>(gedge + (size_t)p * kGraphEdges);
                    e[0] = make_int4(le[0], le[1], le[2], le[3]);
                    e[1] = make_int4(le[4], le[5], le[6], le[7]);
                }
            }
        }
        __syncthreads();
        RS_T(rs_redo);
    }
#undef YP_WL_PUSH
    // kept rows in rank order
    const int max_out = a.max_out > 0 ? a.max_out : 0x7fffffff;
    int nkept = 0;
    for (int i0 = 0; i0 < K && nkept < max_out; i0 += kSegThreads) {
        const int i = i0 + tid;
        const bool kept = i < K && st_get(i) == kKept;
        int total;
        const int rank = tile_rank_rows(kept, s_bal, total);
        if (kept && nkept + rank < max_out) a.kept_slot[reg + nkept + rank] = a.sslot[reg + i];
        nkept += total;
        __syncthreads();
    }
    if (nkept > max_out) nkept = max_out;
    pairs = (unsigned int)wave_sum((int)pairs);
    ties = (unsigned int)wave_sum((int)ties);
    if (lane == 0 && pairs) atomicAdd(&a.seg_pairs[seg], pairs);
    if (lane == 0 && ties) atomicAdd(&a.seg_ties[seg], ties);
    if (tid == 0) a.seg_kept[seg] = nkept;
#ifdef YP_STAMPS
    if (tid == 0 && a.dbg) {
        unsigned long long rs_emit = 0;
        RS_T(rs_emit);
        atomicAdd(a.dbg + 16, rs_rounds);
        atomicAdd(a.dbg + 17, rs_walks);
        atomicAdd(a.dbg + 18, rs_scan);
        atomicAdd(a.dbg + 19, rs_redo);
        atomicAdd(a.dbg + 20, rs_emit);
        atomicAdd(a.dbg + 21, 1ull);
    }
#endif
#undef RS_T
}
