// Host emulation of the few HIP constructs the small bookkeeping kernels use -- TEST
// INFRASTRUCTURE ONLY.  Included in place of <hip/hip_runtime.h> when a kernel source file is
// compiled with g++ for tests/test_v_metrics.py: a launch runs its workgroups one after another,
// every workgroup as blockDim.x OS threads that meet at a std::barrier where the kernel calls
// __syncthreads(), so the kernel's own statements -- indexing, atomics, barriers, float
// arithmetic in source order (-ffp-contract=off) -- execute unchanged on the CPU.  It says nothing
// about timing or about how the GPU schedules the same code; wave-level intrinsics are not provided.
#pragma once
#include <algorithm>
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static /* one workgroup runs at a time, its threads share the variable */
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct alignas(16) float4 {
    float x, y, z, w;
};
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

typedef int hipError_t;
typedef void *hipStream_t;
constexpr hipError_t hipSuccess = 0;
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipMemsetAsync(void *p, int value, size_t n, hipStream_t)
{
    std::memset(p, value, n);
    return hipSuccess;
}

inline thread_local dim3 threadIdx, blockIdx;
inline dim3 blockDim, gridDim;
inline std::barrier<> *emul_barrier = nullptr;
inline void __syncthreads() { emul_barrier->arrive_and_wait(); }

inline int atomicAdd(int *p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline int atomicMin(int *p, int v)
{
    int old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old > v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
    }
    return old;
}

template <class Kernel, class... Args>
void emul_launch(Kernel kernel, dim3 grid, dim3 block, Args... args)
{
    blockDim = block;
    gridDim = grid;
    for (unsigned b = 0; b < grid.x; ++b) {
        std::barrier<> bar((std::ptrdiff_t)block.x);
        emul_barrier = &bar;
        std::vector<std::thread> threads;
        threads.reserve(block.x);
        for (unsigned t = 0; t < block.x; ++t)
            threads.emplace_back([=] {
                threadIdx = dim3(t);
                blockIdx = dim3(b);
                kernel(args...);
                emul_barrier->arrive_and_drop();   // a thread that left early no longer counts
            });
        for (auto &th : threads) th.join();
    }
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) emul_launch(kernel, grid, block, __VA_ARGS__)
