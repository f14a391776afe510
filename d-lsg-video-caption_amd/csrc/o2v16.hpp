// Shared pieces of the second-generation object->frame graph kernels (o2v16.hip: forward, o2v16_bwd.hip: backward):
// tile geometry, the LDS-only barrier and the LDS-DMA piece.  gfx950 only.
#pragma once
#include "common.hpp"
#include "dlsg.h"

namespace o16 {

using namespace dlsg;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

constexpr int O16_THREADS = 512;
constexpr int O16_TILE = 16;

template <int H>
struct O16Geom {
    static constexpr int LDO = H + 4;                          // LDS row stride (floats); a DMA piece never crosses a row
    static constexpr int KW = (H / 16 < 8) ? H / 16 : 8;       // waves that split k in the S product
    static constexpr int HS = H / KW;                          // k slice per wave (multiple of 16)
    static constexpr int NCHUNK = HS / 16;                     // 16-deep chunks = 4 MFMAs per frame block
    static constexpr int NCB = H / 16;                         // 16-column blocks of the aggregation output
    static constexpr int CBW = (NCB + 7) / 8;                  // column blocks per wave
    static constexpr int VB = (H >= 256) ? 16 : 4;             // bytes per lane of one LDS-DMA piece
    static constexpr int PPR = H * 4 / (64 * VB);              // pieces per row
    static constexpr int NP = 2 * PPR;                         // pieces a wave issues per tile (2 rows)
    static constexpr int EPL = H / 64;                         // elements per lane when a wave holds one row
    static constexpr int VEC = (EPL % 4 == 0) ? 4 : 1;
    static constexpr int NCH = EPL / VEC;
    static constexpr int BUF = O16_TILE * LDO;                 // floats per tile buffer
    static constexpr int RED = 8 * 2 * 64 * 4;                 // [wave][frame block][lane] float4
    static constexpr int LDS_FLOATS = 2 * BUF + RED + 2 * H + 64;      // + 2 x (mean[16] | rstd[16]): this tile's and the next's
};

// LDS writes/reads of this wave retired, then the workgroup barrier.  Deliberately NOT __syncthreads(): its fence waits
// vmcnt(0) and would drain the LDS-DMA of the next tile.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// one LDS-DMA piece: 64 lanes x VB bytes from per-lane global addresses to a wave-uniform LDS base (+ lane * VB).
// The 16-byte form is a gfx950 instruction: the host pass of hipcc cannot type-check the builtin (it would silently drop the
// kernel's host stub), so the body exists in the device pass only.
template <int VB>
__device__ __forceinline__ void glds(const char* src, char* dst) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (VB == 16) __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
    else __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)dst, 4, 0, 0);
#endif
}

// The same 16-byte piece with the request written out (wave-uniform row base in SGPRs, per-lane byte offset, LDS byte address
// through m0).  The compiler treats the builtin as an LDS access of unknown extent and puts `s_waitcnt lgkmcnt(0)` (and its
// own vmcnt bookkeeping) around every piece -- between MFMA blocks that is a stall per piece; the kernels order these
// requests themselves (explicit vmcnt before the barrier that publishes a tile).
__device__ __forceinline__ void glds16_asm(const char* base, uint32_t voff, uint32_t lds_addr) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds_addr), "v"(voff), "s"(base) : "memory", "m0");
#endif
}
__device__ __forceinline__ const char* uniform_ptr(const void* q) {
    const uint64_t u = reinterpret_cast<uint64_t>(q);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}

}  // namespace o16
