// Host emulation of the HIP constructs this library's kernels use -- TEST INFRASTRUCTURE ONLY.
//
// Included in place of <hip/hip_runtime.h> when a kernel source file is compiled with g++ for the
// emulation tests.  The kernels' own statements then execute on the CPU: indexing, atomics, LDS
// traffic, barriers, wave collectives and the float arithmetic in source order (-ffp-contract=off),
// inside ordinary host memory where AddressSanitizer sees every access.
//
// Model.  A launch runs its workgroups one after another.  Every thread of a workgroup is a
// fibre (ucontext) on the calling OS thread; a fibre runs until it reaches a collective -- a wave
// intrinsic (__ballot, __shfl*, readlane, readfirstlane, wave_barrier) or __syncthreads -- and
// parks there.  When all live fibres of a wave are parked, the wave's collective is carried out
// and they continue; when all live fibres of the workgroup are parked at __syncthreads, the
// barrier opens.  Fibres of one wave parked at DIFFERENT source lines are in different branches:
// the group at the earliest line goes first, alone, with only its own lanes participating (as the
// hardware's execution mask would have it); a shuffle that reads a non-participating lane returns
// the caller's own value and is counted in emul::stats.
//
// What it is not.  Lanes of a wave do not advance in lockstep between collectives, so code that
// relies on implicit lockstep (LDS exchange inside a wave without a wave_barrier) behaves
// differently here -- such code is wrong on the GPU's memory model too and shows up as a failure.
// Workgroups do not overlap, so kernels whose workgroups wait for each other cannot run.  Timing,
// occupancy, registers, the caches and the scheduler are not modelled at all.
#pragma once
#include <ucontext.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <type_traits>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ thread_local /* one workgroup at a time on one OS thread: its fibres share it */
#define __launch_bounds__(...)
#define __align__(n) __attribute__((aligned(n)))
#define HIP_SYMBOL(x) x

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 {
    float x, y;
};
struct __attribute__((aligned(16))) float4 {
    float x, y, z, w;
};
struct __attribute__((aligned(16))) int4 {
    int x, y, z, w;
};
struct __attribute__((aligned(16))) ulonglong2 {
    unsigned long long x, y;
};
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }

// ------------------------------------------------------------------ host API stand-ins
typedef int hipError_t;
typedef void *hipStream_t;
typedef void *hipEvent_t;
constexpr hipError_t hipSuccess = 0;
constexpr hipError_t hipErrorInvalidValue = 1;
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char *hipGetErrorString(hipError_t) { return "host emulation"; }
inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
inline hipError_t hipMemsetAsync(void *p, int value, size_t n, hipStream_t) { std::memset(p, value, n); return hipSuccess; }
template <class T>
inline hipError_t hipGetSymbolAddress(void **out, T &symbol) { *out = (void *)&symbol; return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }

// ------------------------------------------------------------------ the fibre machine
// Under ThreadSanitizer every fibre is announced to the sanitizer, and every rendezvous is a
// release by all participants followed by an acquire: two threads of a workgroup that touch the
// same LDS or global address, one of them writing, with no barrier or wave collective in between are
// then reported as a data race -- although the fibres run one at a time and the result here may
// well be right.  (Consecutive workgroups and launches are chained by the same means: races BETWEEN
// workgroups are not looked for.)
#if defined(__SANITIZE_THREAD__)
extern "C" {
void *__tsan_create_fiber(unsigned flags);
void __tsan_destroy_fiber(void *fiber);
void __tsan_switch_to_fiber(void *fiber, unsigned flags);   /* flags 1: the switch itself is no synchronisation */
void *__tsan_get_current_fiber(void);
void __tsan_acquire(void *addr);
void __tsan_release(void *addr);
}
#define EMUL_TSAN(...) __VA_ARGS__
#define EMUL_OWN __attribute__((no_sanitize("thread")))   /* the machine's own bookkeeping is not kernel code */
#else
#define EMUL_TSAN(...)
#define EMUL_OWN
#endif

namespace emul {

enum Kind { kNone = 0, kBallot, kShuffle, kFirst, kWaveBarrier, kBlockBarrier, kYield };

struct Lane {
    ucontext_t ctx;
    dim3 tid;
    int index = 0, lane = 0, wave = 0;
    bool done = false;
    int wait = kNone;        // what the fibre is parked at
    int site = 0;            // source line of that collective
    uint64_t post = 0;       // value it contributes
    int src = 0;             // lane a shuffle reads
    uint64_t result = 0;
    void *fiber = nullptr;   // ThreadSanitizer's handle
};

struct Stats {
    long collectives = 0, divergent_groups = 0, shuffle_from_inactive = 0, launches = 0, blocks = 0;
};

inline Stats stats;
inline bool trace_divergence = std::getenv("YP_EMUL_TRACE") != nullptr;
inline Lane *cur = nullptr;
inline dim3 g_block_idx, g_block_dim, g_grid_dim;
inline ucontext_t sched_ctx;
inline std::vector<Lane> lanes;
inline std::vector<char *> stacks;
inline const std::function<void()> *body = nullptr;
constexpr size_t kStackBytes = 256 << 10;
inline void *sched_fiber = nullptr;
inline char wave_token[16], block_token, begin_token, end_token;     // addresses the sanitizer hangs its clocks on
// (YP_EMUL_TSAN_SELFTEST=1 hides the workgroup barriers from the sanitizer -- every kernel that needs
// its barriers must then be reported: the check that the race detection is alive)
inline const bool tsan_selftest = std::getenv("YP_EMUL_TSAN_SELFTEST") != nullptr;
inline char dummy_token[1024];
inline void *token_of(int kind, int wave)
{
    if (kind == kBlockBarrier) return tsan_selftest ? (void *)&dummy_token[cur->index & 1023] : (void *)&block_token;
    return (void *)&wave_token[wave & 15];
}

EMUL_OWN inline void fibre_entry()
{
    EMUL_TSAN(__tsan_acquire(&begin_token);)      // what the host and earlier workgroups did comes first
    (*body)();
    cur->done = true;
    EMUL_TSAN(__tsan_release(&end_token); __tsan_switch_to_fiber(sched_fiber, 1);)
    swapcontext(&cur->ctx, &sched_ctx);
}

// Parks the running fibre at a collective and returns its result once the scheduler has
// carried the collective out.
EMUL_OWN inline uint64_t park(int kind, int site, uint64_t post, int src = 0)
{
    Lane *me = cur;
    me->wait = kind;
    me->site = site;
    me->post = post;
    me->src = src;
    EMUL_TSAN(__tsan_release(token_of(kind, me->wave)); __tsan_switch_to_fiber(sched_fiber, 1);)
    swapcontext(&me->ctx, &sched_ctx);
    EMUL_TSAN(__tsan_acquire(token_of(kind, me->wave));)
    return me->result;
}

EMUL_OWN [[noreturn]] inline void die(const char *what)
{
    std::fprintf(stderr, "host emulation: %s (block %u,%u,%u)\n", what, g_block_idx.x, g_block_idx.y, g_block_idx.z);
    for (const Lane &l : lanes)
        if (!l.done) std::fprintf(stderr, "  thread %d: wait kind %d at line %d\n", l.index, l.wait, l.site / 64);
    std::abort();
}

// Carries out what the parked fibres of one wave allow.  Returns true if any fibre was released.
EMUL_OWN inline bool resolve_wave(int w, int n_threads)
{
    const int lo = w * 64, hi = std::min(n_threads, lo + 64);
    int site = 0x7fffffff, kind = kNone;
    for (int i = lo; i < hi; ++i) {
        const Lane &l = lanes[i];
        if (l.done) continue;
        if (l.wait == kNone || l.wait == kYield) return false;  // somebody can still run
        if (l.site < site) { site = l.site; kind = l.wait; }
    }
    if (kind == kNone || kind == kBlockBarrier) return false;  // (the workgroup barrier is resolved by the caller)
    uint64_t ballot = 0;
    int first = -1, others = 0;
    for (int i = lo; i < hi; ++i) {
        const Lane &l = lanes[i];
        if (l.done) continue;
        if (l.site != site) { ++others; continue; }
        if (l.wait != kind) die("two kinds of collective at one source line");
        if (first < 0) first = i;
        if (kind == kBallot && l.post) ballot |= 1ull << l.lane;
    }
    if (others) {
        ++stats.divergent_groups;
        if (trace_divergence) {
            int other_site = 0;
            for (int i = lo; i < hi; ++i)
                if (!lanes[i].done && lanes[i].site != site) other_site = lanes[i].site;
            std::fprintf(stderr, "host emulation: wave %d split: line %d (kind %d) goes before line %d; %d lanes wait\n",
                         w, site / 64, kind, other_site / 64, others);
        }
    }
    ++stats.collectives;
    for (int i = lo; i < hi; ++i) {
        Lane &l = lanes[i];
        if (l.done || l.site != site) continue;
        switch (kind) {
        case kBallot: l.result = ballot; break;
        case kFirst: l.result = lanes[first].post; break;
        case kShuffle: {
            const int s = lo + l.src;
            if (l.src >= 0 && s < hi && !lanes[s].done && lanes[s].site == site && lanes[s].wait == kShuffle) {
                l.result = lanes[s].post;
            } else {
                l.result = l.post;     // outside the wave: the defined result; a lane that is not taking part: undefined on the GPU
                if (l.src >= 0 && l.src < 64) ++stats.shuffle_from_inactive;
            }
            break;
        }
        default: l.result = 0; break;
        }
    }
    for (int i = lo; i < hi; ++i) {
        Lane &l = lanes[i];
        if (!l.done && l.site == site && l.wait == kind) l.wait = kNone, l.site = 0;
    }
    return true;
}

EMUL_OWN inline void run_block(int n_threads)
{
    if ((int)lanes.size() < n_threads) lanes.resize(n_threads);
    while ((int)stacks.size() < n_threads) stacks.push_back((char *)std::malloc(kStackBytes));
    for (int i = 0; i < n_threads; ++i) {
        Lane &l = lanes[i];
        l.tid = dim3((unsigned)i);
        l.index = i;
        l.lane = i & 63;
        l.wave = i >> 6;
        l.done = false;
        l.wait = kNone;
        l.site = 0;
        getcontext(&l.ctx);
        l.ctx.uc_stack.ss_sp = stacks[i];
        l.ctx.uc_stack.ss_size = kStackBytes;
        l.ctx.uc_link = &sched_ctx;
        makecontext(&l.ctx, fibre_entry, 0);
        EMUL_TSAN(l.fiber = __tsan_create_fiber(0);)
    }
    EMUL_TSAN(sched_fiber = __tsan_get_current_fiber(); __tsan_release(&begin_token);)
    const int n_waves = (n_threads + 63) / 64;
    for (;;) {
        bool progress = false, all_done = true;
        for (int w = 0; w < n_waves; ++w) {
            bool again = true;
            while (again) {
                again = false;
                for (int i = w * 64; i < std::min(n_threads, w * 64 + 64); ++i) {
                    Lane &l = lanes[i];
                    if (l.wait == kYield) l.wait = kNone;
                    if (l.done || l.wait != kNone) continue;
                    cur = &l;
                    EMUL_TSAN(__tsan_switch_to_fiber(l.fiber, 1);)
                    swapcontext(&sched_ctx, &l.ctx);
                    progress = true;
                }
                for (int i = w * 64; i < std::min(n_threads, w * 64 + 64); ++i)
                    if (!lanes[i].done && lanes[i].wait == kYield) again = true;
                if (!again && resolve_wave(w, n_threads)) again = progress = true;
            }
        }
        bool at_barrier = true;
        for (int i = 0; i < n_threads; ++i) {
            if (lanes[i].done) continue;
            all_done = false;
            if (lanes[i].wait != kBlockBarrier) at_barrier = false;
        }
        if (all_done) break;
        if (at_barrier) {
            for (int i = 0; i < n_threads; ++i)
                if (!lanes[i].done) lanes[i].wait = kNone, lanes[i].site = 0;
            ++stats.collectives;
            continue;
        }
        if (!progress) die("deadlock: no fibre can run and no collective is complete");
    }
    cur = nullptr;
    EMUL_TSAN(__tsan_acquire(&end_token); for (int i = 0; i < n_threads; ++i) __tsan_destroy_fiber(lanes[i].fiber);)
}

inline const int block_order = [] {
    const char *e = std::getenv("YP_EMUL_ORDER");
    return !e ? 0 : (std::strcmp(e, "reverse") == 0 ? 1 : (std::strcmp(e, "shuffle") == 0 ? 2 : 0));
}();

// A permutation of 0..n-1: multiplication by an odd constant modulo the next power of two,
// walked until it lands inside the range (cycle walking).
EMUL_OWN inline unsigned long scrambled(unsigned long i, unsigned long n)
{
    unsigned long m = 1;
    while (m < n) m <<= 1;
    unsigned long v = i;
    do v = (v * 2654435761ul + 40503ul) & (m - 1); while (v >= n);
    return v;
}

EMUL_OWN inline void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()> &kernel_call)
{
    if (block.y != 1 || block.z != 1) die("only one-dimensional workgroups are emulated");
    if (block.x == 0 || block.x > 1024 || lds_bytes > (160u << 10)) die("launch shape beyond the hardware's");
    if (cur != nullptr) die("launch from inside a kernel");
    body = &kernel_call;
    g_block_dim = block;
    g_grid_dim = grid;
    ++stats.launches;
    // Workgroups of a launch may run in any order on the GPU: YP_EMUL_ORDER=reverse runs them last
    // to first, =shuffle in a scrambled order, to show that a result does not lean on the order.
    const unsigned long n_blocks = (unsigned long)grid.x * grid.y * grid.z;
    for (unsigned long n = 0; n < n_blocks; ++n) {
        unsigned long b = n;
        if (block_order == 1) b = n_blocks - 1 - n;
        if (block_order == 2) b = scrambled(n, n_blocks);
        g_block_idx = dim3((unsigned)(b % grid.x), (unsigned)(b / grid.x % grid.y), (unsigned)(b / grid.x / grid.y));
        ++stats.blocks;
        run_block((int)block.x);
    }
    body = nullptr;
}

template <class T>
inline uint64_t bits_of(T v)
{
    static_assert(sizeof(T) <= 8, "shuffles move at most 64 bits");
    uint64_t b = 0;
    std::memcpy(&b, &v, sizeof(T));
    return b;
}
template <class T>
inline T from_bits(uint64_t b)
{
    T v;
    std::memcpy(&v, &b, sizeof(T));
    return v;
}
template <class T>
EMUL_OWN inline T shuffle(T v, int src, int site)
{
    return from_bits<T>(park(kShuffle, site, bits_of(v), src));
}

template <class T>
EMUL_OWN inline T peek_lane(const T *mine, int l)
{
    // let every other fibre that can run reach its next parking place first: the lane being read
    // may still be on its way from the last collective to the statement that sets the variable
    park(kYield, 0, 0);
    const Lane *me = cur;
    const long off = reinterpret_cast<const char *>(mine) - stacks[me->index];
    if (off < 0 || off + (long)sizeof(T) > (long)kStackBytes) die("readlane of something that is not a local variable");
    if (l < 0 || l > 63 || me->wave * 64 + l >= (int)g_block_dim.x) die("readlane of a lane outside the wave");
    T v;
    std::memcpy(&v, stacks[me->wave * 64 + l] + off, sizeof(T));
    return v;
}

template <class T>
inline std::remove_const_t<T> read_lane(T &v, int l, int)
{
    return peek_lane<std::remove_const_t<T>>(&v, l);
}
template <class T, class = std::enable_if_t<!std::is_lvalue_reference_v<T>>>
inline T read_lane(T &&v, int l, int site)
{
    return shuffle<T>(v, l, site);
}

template <class T>
EMUL_OWN inline T first_lane(T v, int site)
{
    return from_bits<T>(park(kFirst, site, bits_of(v)));
}

}  // namespace emul

#define threadIdx (emul::cur->tid)
#define blockIdx (emul::g_block_idx)
#define blockDim (emul::g_block_dim)
#define gridDim (emul::g_grid_dim)

#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
    emul::launch(grid, block, lds, [&] { kernel(__VA_ARGS__); })
#define hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, ev0, ev1, flags, ...) \
    emul::launch(grid, block, lds, [&] { kernel(__VA_ARGS__); })

// ------------------------------------------------------------------ device intrinsics
#define EMUL_SITE (__LINE__ * 64 + (int)(sizeof(__FILE__) % 64))
#define __syncthreads() ((void)emul::park(emul::kBlockBarrier, EMUL_SITE, 0))
#define __threadfence_block() ((void)0)
#define __threadfence() ((void)0)
#define __ballot(p) ((unsigned long long)emul::park(emul::kBallot, EMUL_SITE, (p) ? 1 : 0))
#define __any(p) (emul::park(emul::kBallot, EMUL_SITE, (p) ? 1 : 0) != 0)
#define __shfl(v, src, ...) emul::shuffle((v), (int)(src), EMUL_SITE)
#define __shfl_xor(v, mask, ...) emul::shuffle((v), emul::cur->lane ^ (int)(mask), EMUL_SITE)
#define __shfl_up(v, delta, ...) emul::shuffle((v), emul::cur->lane - (int)(delta), EMUL_SITE)
#define __shfl_down(v, delta, ...) emul::shuffle((v), emul::cur->lane + (int)(delta), EMUL_SITE)
// v_readlane_b32 reads a lane's register whatever the execution mask says, also from inside a
// branch the lane did not take.  Here: the same local variable in the other fibre's frame (the
// fibres of a wave run the same function at the same stack depth, so the variable sits at the same
// offset of every fibre's stack).  The argument therefore has to be a named local variable that the
// lane wrote before the wave's last collective; any other expression is read as a shuffle among
// the lanes that take part.
#define __builtin_amdgcn_readlane(v, l) emul::read_lane((v), (int)(l), EMUL_SITE)
#define __builtin_amdgcn_readfirstlane(v) emul::first_lane((v), EMUL_SITE)
// v_writelane_b32 (inline assembly on the device): a value that is uniform over the wave goes into
// ONE lane of a register
#define YP_WRITELANE(dst, val, l) ((dst) = (emul::cur->lane == (l)) ? (val) : (dst))
#define __builtin_amdgcn_wave_barrier() ((void)emul::park(emul::kWaveBarrier, EMUL_SITE, 0))
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_s_sleep(n) ((void)0)

EMUL_OWN inline unsigned emul_lanes_below(unsigned lo_or_hi_mask, bool hi)
{
    const int lane = emul::cur->lane;
    unsigned below;
    if (!hi) below = lane >= 32 ? 0xffffffffu : ((1u << lane) - 1u);
    else below = lane < 32 ? 0u : ((1u << (lane - 32)) - 1u);
    return (unsigned)__builtin_popcount(lo_or_hi_mask & below);
}
#define __builtin_amdgcn_mbcnt_lo(mask, base) ((base) + emul_lanes_below((mask), false))
#define __builtin_amdgcn_mbcnt_hi(mask, base) ((base) + emul_lanes_below((mask), true))

#define __popcll(v) __builtin_popcountll(v)
#define __popc(v) __builtin_popcount(v)
#define __ffsll(v) __builtin_ffsll(v)
inline unsigned __float_as_uint(float f) { return emul::from_bits<unsigned>(emul::bits_of(f)); }
inline int __float_as_int(float f) { return emul::from_bits<int>(emul::bits_of(f)); }
inline float __uint_as_float(unsigned u) { return emul::from_bits<float>(emul::bits_of(u)); }
inline float __int_as_float(int i) { return emul::from_bits<float>(emul::bits_of(i)); }

// Atomics: the fibres of a workgroup share one OS thread and workgroups do not overlap, so any
// read-modify-write is atomic here; integers go through the compiler's atomic builtins all the same
// so that ThreadSanitizer (emul_driver_tsan) knows them for what they are.
template <class T>
inline T atomicAdd(T *p, T v)
{
    if constexpr (std::is_integral_v<T>) return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
    else { const T old = *p; *p = old + v; return old; }
}
inline int atomicAdd(int *p, unsigned v) { return __atomic_fetch_add(p, (int)v, __ATOMIC_RELAXED); }
template <class T>
inline T atomicMin(T *p, T v)
{
    T old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v < old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
    }
    return old;
}
template <class T>
inline T atomicMax(T *p, T v)
{
    T old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v > old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
    }
    return old;
}
template <class T>
inline T atomicOr(T *p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }

inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }

// Single-instruction float max / min of the kernels (inline GCN assembly on the device).
#define YP_VMAX_F32(r, a, b) ((r) = std::fmax((a), (b)))
#define YP_VMIN_F32(r, a, b) ((r) = std::fmin((a), (b)))
