// Building blocks of the persistent recurrent kernels (bilstm.hip: EncoderVisual's BiLSTM; critic_lstm.hip: DiscV2's LSTM at three
// differentiation levels): write-through exchange of a (rows x H) state between the workgroups of one launch through L2, flags,
// the LDS image of a 32-column weight slice and the 64 x 32 MFMA product whose A operand streams straight from the exchange
// buffer into registers.  See bilstm.hip for the protocol (MI355X_MICROARCH.md, inter-workgroup visibility) and the layouts.
#pragma once
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace persist {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int THREADS = 256;            // 4 waves, one per SIMD
constexpr int ROWS = 64;                // rows of a workgroup = positions of an exchange slot
constexpr unsigned SPIN_LIMIT = 1u << 21;
constexpr int SC1 = 16;                 // aux bit of the raw buffer builtins: sc1 (write-through store / L1-bypassing load)

__device__ __forceinline__ f32x4 ld_sc1(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, SC1));
}
__device__ __forceinline__ void st_sc1(__amdgpu_buffer_rsrc_t r, int byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, SC1);
}
__device__ __forceinline__ f32x2 ld2_sc1(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, SC1));
}
__device__ __forceinline__ void st2_sc1(__amdgpu_buffer_rsrc_t r, int byte_off, f32x2 v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, byte_off, 0, SC1);
}

// wave-level wait: lanes [0, n) each watch one flag word until all of them read `want`.  Returns false on time-out.
__device__ __forceinline__ bool wait_flags(const uint32_t* flags, int n, uint32_t want) {
    const int lane = threadIdx.x & 63;
    for (unsigned spins = 0;; ++spins) {
        uint32_t v = want;
        if (lane < n) v = __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all(v == want)) break;
        if (spins > SPIN_LIMIT) return false;
        __builtin_amdgcn_s_sleep(4);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      // compiler-only: the payload loads stay below the poll
    return true;
}
// every storing wave drains its write-through stores, the workgroup meets, one lane raises the flag
__device__ __forceinline__ void publish(uint32_t* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// LDS image of a 32-column weight slice as the B operand of v_mfma_f32_16x16x4_f32 (ds_read_b128, conflict-free):
//   slot(cb, jb, kg, c) = ((cb * NJB + jb) * 4 + kg) * 16 + c  holds k = 16 jb + 4 kg + (0..3) of column (cb, c).
// "NT" form (y = x W^T): column (cb, c) = row `rowof(cb, c)` of W (row-major, k contiguous).
template <int J, class RowOf>
__device__ __forceinline__ void fill_wimage_rows(f32x4* wimg, const float* W, int ldw, RowOf rowof) {
    constexpr int NJB = 4 * J;
    f32x4 tmp[2 * J];
#pragma unroll
    for (int it = 0; it < 2 * J; ++it) {
        const int s = threadIdx.x + it * THREADS;
        const int cc = s & 15, kg = (s >> 4) & 3, jb = (s >> 6) % NJB, cb = (s >> 6) / NJB;
        tmp[it] = *reinterpret_cast<const f32x4*>(W + (int64_t)rowof(cb, cc) * ldw + 16 * jb + 4 * kg);
    }
#pragma unroll
    for (int it = 0; it < 2 * J; ++it) wimg[threadIdx.x + it * THREADS] = tmp[it];
}
// "NN" form (y = x W): column (cb, c) = column col0 + 16 cb + c of W, k = row k0 + .. of W (row-major).
template <int J>
__device__ __forceinline__ void fill_wimage_cols(f32x4* wimg, const float* W, int ldw, int k0, int col0) {
    constexpr int NJB = 4 * J;
#pragma unroll 4
    for (int it = 0; it < 2 * J; ++it) {
        const int s = threadIdx.x + it * THREADS;
        const int cc = s & 15, kg = (s >> 4) & 3, jb = (s >> 6) % NJB, cb = (s >> 6) / NJB;
        const float* src = W + (int64_t)(k0 + 16 * jb + 4 * kg) * ldw + col0 + 16 * cb + cc;
        wimg[s] = f32x4{src[0], src[ldw], src[2 * ldw], src[3 * ldw]};
    }
}

// own[i][cb] <- the 64 x 32 product of the exchange slot at byte offset `slot_base` (k-major [k][64 rows], k in [0, 64 J))
// with the weight image: wave w contracts k in [16 J w, 16 J (w + 1)), the four K-partials are summed through `red`
// (4 * 3 * 8 * 64 floats) so that wave w ends with accumulator register w of every block:
//   own[i][cb] = value at row 16 q + 4 w + i, column (cb, c)        (lane = 16 q + c)
// All 4 J loads of a wave are issued before its first MFMA.  Contains one __syncthreads().
template <int J>
__device__ __forceinline__ void product_64x32(__amdgpu_buffer_rsrc_t xbuf, int slot_base, const f32x4* wimg, float* red,
                                              float (&own)[4][2]) {
    constexpr int NJB = 4 * J;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int abase = slot_base + ((w * (16 * J) + 4 * q) * ROWS + 4 * c) * 4;
    f32x4 av[J][4];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int s = 0; s < 4; ++s) av[j][s] = ld_sc1(xbuf, abase + (16 * j + s) * ROWS * 4);
    __builtin_amdgcn_sched_barrier(0);      // the scheduler must not sink the loads next to their uses (one L2 round trip each)
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = acc[i][0]; }
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int jb = w * J + j;
        const f32x4 b0 = wimg[(jb * 4 + q) * 16 + c];
        const f32x4 b1 = wimg[((NJB + jb) * 4 + q) * 16 + c];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][s][i], b0[s], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][s][i], b1[s], acc[i][1], 0, 0, 0);
            }
    }
#pragma unroll
    for (int dst = 0; dst < 4; ++dst)
        if (dst != w) {
            float* p = red + ((dst * 3 + (w - (w > dst))) * 8) * 64 + lane;
#pragma unroll
            for (int i = 0; i < 4; ++i) { p[(2 * i) * 64] = acc[i][0][dst]; p[(2 * i + 1) * 64] = acc[i][1][dst]; }
        }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (r == w) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { own[i][0] = acc[i][0][r]; own[i][1] = acc[i][1][r]; }
        }
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
        const float* p = red + ((w * 3 + s3) * 8) * 64 + lane;
#pragma unroll
        for (int i = 0; i < 4; ++i) { own[i][0] += p[(2 * i) * 64]; own[i][1] += p[(2 * i + 1) * 64]; }
    }
}

inline int device_cus() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return cus;
}

}  // namespace persist
