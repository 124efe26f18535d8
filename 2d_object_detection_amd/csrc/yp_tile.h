// yp_tile.h -- coalesced access to row-major records for the thread-per-row kernels.
//
// A kernel whose thread t owns record t of a tile (W consecutive floats, e.g. the 5+C values
// of one (cell, anchor) slot) would read its record with a stride of W floats between lanes.
// Instead the workgroup copies the tile's kTileRows*W contiguous floats through LDS with
// lane-consecutive 4-byte accesses (256 B per wave instruction) and every thread then takes
// its record from LDS; results that are records themselves go back the same way.  Records
// wider than kTileMaxWidth floats do not fit the LDS budget and use the direct path.
#pragma once
#include <hip/hip_runtime.h>

constexpr int kTileRows = 256;      // records per tile == threads per workgroup
constexpr int kTileMaxWidth = 16;   // widest record staged through LDS (16 KB per tile)

// LDS bytes a kernel of this family needs for records of `width` floats.
inline size_t yp_tile_lds_bytes(int width)
{
    return width <= kTileMaxWidth ? (size_t)kTileRows * width * sizeof(float) : 0;
}

// Copies the tile starting at src_tile (valid_floats of it exist) into lds and ends with a
// barrier; afterwards record t is lds[t*width .. t*width+width).
__device__ __forceinline__ void yp_tile_load(const float *__restrict__ src_tile, long long valid_floats,
                                             int width, float *lds)
{
    const int total = kTileRows * width;
    for (int e = threadIdx.x; e < total; e += kTileRows) lds[e] = e < valid_floats ? src_tile[e] : 0.0f;
    __syncthreads();
}

// Writes the records the threads left in lds (record t at lds[t*width]) to dst_tile; starts
// with a barrier (so every thread's record is in place).
__device__ __forceinline__ void yp_tile_store(float *__restrict__ dst_tile, long long valid_floats, int width,
                                              const float *lds)
{
    __syncthreads();
    const int total = kTileRows * width;
    for (int e = threadIdx.x; e < total; e += kTileRows)
        if (e < valid_floats) dst_tile[e] = lds[e];
}
