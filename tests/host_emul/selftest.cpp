// Known answers for the emulation itself: what every collective returns, in uniform and in
// divergent control flow, the order barriers impose, and the bookkeeping of the counters.
#include <hip/hip_runtime.h>

#include <cstdio>

static int g_fail = 0, g_checks = 0;
#define CHECK(c) do { ++g_checks; if (!(c)) { ++g_fail; std::printf("FAILED %s:%d %s\n", __FILE__, __LINE__, #c); } } while (0)

__global__ void k_collectives(unsigned long long *ballots, int *vals)
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const unsigned long long even = __ballot((lane & 1) == 0);
    if (lane == 0) ballots[wave] = even;
    vals[t * 8 + 0] = __shfl_xor(lane * 10, 1, 64);
    vals[t * 8 + 1] = __shfl_up(lane * 10, 2, 64);
    vals[t * 8 + 2] = __shfl_down(lane * 10, 3, 64);
    vals[t * 8 + 3] = __shfl(lane * 10, 5, 64);
    int mine = 1000 * wave + lane;
    __builtin_amdgcn_wave_barrier();
    vals[t * 8 + 4] = __builtin_amdgcn_readlane(mine, 7);
    vals[t * 8 + 5] = __builtin_amdgcn_readfirstlane(mine);
    const unsigned lo = (unsigned)even, hi = (unsigned)(even >> 32);
    vals[t * 8 + 6] = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
    // a ballot inside a branch only sees the lanes that took the branch
    unsigned long long inner = 0;
    if (lane >= 48) inner = __ballot(true);
    // ... and readlane still reads a lane that did not take it
    int peek = -1;
    if (lane < 4) peek = __builtin_amdgcn_readlane(mine, 60);
    // everybody meets again at the next collective: the lanes that skipped the branch wait there
    // while the branch's ballot is carried out (counted as a divergent group)
    const unsigned long long all = __ballot(true);
    vals[t * 8 + 7] = all != ~0ull ? -7 : (lane >= 48 ? (int)(inner >> 48) : peek);
}

__global__ void k_barrier_order(int *out)
{
    __shared__ int s_v[256];
    const int t = threadIdx.x;
    s_v[t] = t * t;
    __syncthreads();
    const int other = s_v[255 - t];           // written by a thread that runs later in the emulation
    __syncthreads();
    s_v[t] = other + 1;
    __syncthreads();
    out[blockIdx.x * 256 + t] = s_v[(t + 64) & 255] + (int)blockIdx.x + (int)gridDim.x * 1000;
}

__global__ void k_atomics(int *counter, int *low, unsigned *bits)
{
    __shared__ int s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    atomicAdd(&s_n, 2);
    atomicMin(low, (int)(blockIdx.x * 1000 + threadIdx.x) - 5);
    atomicOr(bits, 1u << (threadIdx.x & 31));
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(counter, s_n);
    if (threadIdx.x >= 100) return;           // threads that leave early do not hold up a barrier
    __syncthreads();
    if (threadIdx.x == 1) atomicAdd(counter, 1);
}

int main()
{
    static unsigned long long ballots[2];
    static int vals[128 * 8];
    hipLaunchKernelGGL(k_collectives, dim3(1), dim3(128), 0, nullptr, ballots, vals);
    CHECK(ballots[0] == 0x5555555555555555ull && ballots[1] == 0x5555555555555555ull);
    for (int t = 0; t < 128; ++t) {
        const int lane = t & 63, wave = t >> 6;
        const int *v = vals + t * 8;
        CHECK(v[0] == (lane ^ 1) * 10);
        CHECK(v[1] == (lane >= 2 ? lane - 2 : lane) * 10);
        CHECK(v[2] == (lane + 3 < 64 ? lane + 3 : lane) * 10);
        CHECK(v[3] == 50);
        CHECK(v[4] == 1000 * wave + 7);
        CHECK(v[5] == 1000 * wave);
        CHECK(v[6] == (lane + 1) / 2);
        CHECK(v[7] == (lane >= 48 ? 0xffff : (lane < 4 ? 1000 * wave + 60 : -1)));
    }
    CHECK(emul::stats.divergent_groups > 0);            // the branch above was seen as one
    CHECK(emul::stats.shuffle_from_inactive == 0);

    static int out[3 * 256];
    hipLaunchKernelGGL(k_barrier_order, dim3(3), dim3(256), 0, nullptr, out);
    for (int b = 0; b < 3; ++b)
        for (int t = 0; t < 256; ++t) {
            const int u = (t + 64) & 255;
            CHECK(out[b * 256 + t] == (255 - u) * (255 - u) + 1 + b + 3000);
        }

    int counter = 0, low = 1 << 30;
    unsigned bits = 0;
    hipLaunchKernelGGL(k_atomics, dim3(4), dim3(192), 0, nullptr, &counter, &low, &bits);
    CHECK(counter == 4 * (2 * 192 + 1) && low == -5 && bits == 0xffffffffu);
    CHECK(emul::stats.launches == 3 && emul::stats.blocks == 1 + 3 + 4);

    if (g_fail) { std::printf("%d of %d checks failed\n", g_fail, g_checks); return 1; }
    std::printf("ok %d\n", g_checks);
    return 0;
}
