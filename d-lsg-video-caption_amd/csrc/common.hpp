// Shared device helpers for the D-LSG hot-path kernels (gfx950 / CDNA4 only, wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dlsg.h"   // DLSG_OK / DLSG_E* return codes are part of the public ABI

#define DLSG_CHECK_LAUNCH()                                  \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return DLSG_ELAUNCH;          \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace dlsg {

// Wave-wide reductions without the LDS crossbar: four DPP steps inside each row of 16 lanes (quad swaps, half-row and row
// mirror), then the four row results through scalar registers (v_readlane).  ~10 VALU issue slots instead of six dependent
// ds_bpermute round trips (~100+ cycles each, and they queue behind the workgroup's other LDS traffic); every lane gets the
// result.  (`__shfl_xor` trees measured 5.0k of 19.4k cycles per tile in the object->frame kernel, tools/o2v_stamps.py.)
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum_dpp(float v) {
    v += dpp_f32<0xB1>(v);       // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);       // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(v);      // row_half_mirror
    v += dpp_f32<0x140>(v);      // row_mirror
    return v;
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v = row16_sum_dpp(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float wave_sum(float v) { return wave_sum_dpp(v); }
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f32<0xB1>(v));
    v = fmaxf(v, dpp_f32<0x4E>(v));
    v = fmaxf(v, dpp_f32<0x141>(v));
    v = fmaxf(v, dpp_f32<0x140>(v));
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// Block-wide sum over blockDim.x threads (multiple of 64, <= 1024).  `red` = >= 16 floats of LDS.
// Every thread gets the result.  Two barriers; safe to call repeatedly with the same `red`.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += red[i];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < nw; ++i) r = fmaxf(r, red[i]);
    return r;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// Counter-based dropout mask: keep iff hash(seed, site, idx) >= p.  Stateless, so the backward pass
// recomputes the same mask from (seed, site, idx) instead of storing it.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float drop_scale(uint64_t seed, uint32_t site, uint64_t idx, float p) {
    // returns 0 (dropped) or 1/(1-p) (kept)
    uint32_t h = mix32((uint32_t)idx ^ mix32((uint32_t)(idx >> 32) + 0x9e3779b9U * site + (uint32_t)seed));
    h = mix32(h ^ (uint32_t)(seed >> 32));
    const float u = (float)(h >> 8) * (1.0f / 16777216.0f);
    return u < p ? 0.f : 1.f / (1.f - p);
}

// XCD-aware workgroup -> (row tile, group, column tile) map of the grouped GEMMs, grid = (tiles_m * tiles_n, groups).
// Workgroups are dealt round-robin over the 8 XCDs in dispatch order (x fastest, then y) and each XCD has its own 4-MB L2:
//   * a multiple of 8 groups (the deep weight gradients: 8 row chunks x streams): a whole GROUP per XCD -- its A and B
//     panels are fetched into one L2 instead of all eight (the 26624-deep obj_embed gradient read 4.9x its algorithmic
//     bytes at the L2 -> fabric boundary when every XCD walked every B panel of every group; 2.1x with this map);
//   * otherwise, per group, the blocks an XCD receives (blockIdx.x % 8) walk consecutive column tiles of one A row panel.
//     (Tried and dropped: interleaving the groups of a launch so that the two streams' region projections, which read
//     the same 218 MB of regions, sit next to each other -- with half as many row panels in flight per XCD the weight
//     panels were re-fetched more often than the shared A saved: 2.1 GB vs 1.3-1.5 GB per launch, FETCH_SIZE.)
// Bijective for any grid; locality is a speed matter only.
__device__ __forceinline__ void gemm_tile_map(int tiles_n, int& tm, int& z, int& tn) {
    const int nblk = gridDim.x, Z = gridDim.y;
    int t;
    if ((Z & 7) == 0) {
        const int lin = blockIdx.x + nblk * blockIdx.y;
        const int xcd = lin & 7, j = lin >> 3;
        z = xcd + 8 * (j / nblk);
        t = j % nblk;
    } else {
        const int bid = blockIdx.x, xcd = bid & 7, q = nblk >> 3, rmd = nblk & 7;
        t = (xcd < rmd ? xcd * (q + 1) : rmd * (q + 1) + (xcd - rmd) * q) + (bid >> 3);
        z = blockIdx.y;
    }
    tm = t / tiles_n; tn = t % tiles_n;
}

}  // namespace dlsg
