// Large-tile fp32 GEMM for the products that fill the chip many times over (region projections, their input and weight
// gradients): 256 x 256 or 256 x 128 output tile per workgroup on v_mfma_f32_32x32x2_f32, exact fp32 (the same k-ordered fmaf
// chain per output element as gemm.hip).
//
// Why a second tile family: the 128 x 128 kernel of gemm.hip stages 32 flop per operand byte through LDS with two barriers
// per 32-deep K tile and leans on three workgroups per CU to cover them; on the step's largest launches it runs at 111-122
// TFLOP/s where the vendor library's 256 x 256 kernel reaches 136-138 (tools/archive/gemm_vs_rocblas.py).  Here:
//   * 4 waves as 2 x 2, each wave a 128 x 128 (or 128 x 64) block of the tile = 16 (8) accumulators of 32 x 32: one LDS
//     fragment feeds four MFMAs, 64 flop per staged byte;
//   * stage = 32 k of both operands, filled by LDS-DMA (`global_load_lds_dwordx4`, no staging registers, no ds_write) into
//     one of two LDS buffers while the other one is multiplied: ONE barrier per stage, 256 (128) MFMAs per wave between barriers;
//   * k-contiguous operands: rows x 128 B, lane-linear 1-KB pieces of 8 rows, swizzled on the SOURCE address (slot (row, s) holds
//     k-segment s ^ ((row >> 1) & 7)): b128 fragment reads with lane = row are conflict-free in the hardware's 16-lane groups;
//     row-contiguous operands (the TN form's both, the NN form's B): [k][rows], read by b32 with lanes on consecutive banks;
//   * lane (r, h) feeds k = 4 (2q + h) + j for MFMA j of k-group q: A and B use the same k order;
//   * the next k-group's fragments are read before the current group's MFMAs are issued (double-buffered registers), the
//     next stage's DMA pieces are issued a few at a time between the groups.
// Needs 16-B aligned operands, strides and (for row-contiguous operands) widths that are multiples of 4, K >= 32, K % 4 == 0;
// the dispatcher (gemm.hip) sends everything else to the 128 / 64 tiles.
#include <mutex>

#include "common.hpp"
#include "dlsg.h"

namespace {

constexpr int NT = 256;
constexpr int BK = 32;

struct KArgs {
    int M, N, ldc, ngroups, flags;
    int64_t bsa, bsb, bsc;
    float alpha;
    const float* bias;
    const int32_t* skip_if;
    dlsg_gemm_group g[DLSG_GEMM_MAXG];
};

__device__ __forceinline__ void glds16(const float* src, char* dst) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) void* lp_t;
    typedef const __attribute__((address_space(1))) void* gp_t;
    __builtin_amdgcn_global_load_lds((gp_t)src, (lp_t)dst, 16, 0, 0);
#endif
}

// DMA piece `pc` (1 KB) of an operand image: ROWS x 32 floats of the K range [koff, koff + 32).
//   T == false: element (row, k) at base[row * ld + k]   -> image [row][32], swizzled 16-B segments
//   T == true : element (row, k) at base[k * ld + row]   -> image [k][ROWS]
template <int ROWS, bool T>
__device__ __forceinline__ void issue_piece(const float* __restrict__ base, int64_t ld, int row0, int rmax, int koff, int pc,
                                            int lane, char* img) {
    if (!T) {
        const int R = 8 * pc + (lane >> 3);
        const int seg = (lane & 7) ^ ((R >> 1) & 7);
        glds16(base + (int64_t)min(row0 + R, rmax - 1) * ld + koff + 4 * seg, img + pc * 1024);
    } else {
        constexpr int KR = 256 / ROWS;                     // k-rows per piece
        constexpr int LPR = 64 / KR;                       // lanes per k-row
        const int kr = KR * pc + lane / LPR;
        const int c = min(row0 + 4 * (lane % LPR), rmax - 4);      // (columns past the edge are never stored)
        glds16(base + (int64_t)(koff + kr) * ld + c, img + pc * 1024);
    }
}

template <int BM, int BN, bool AT, bool BT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_big_kernel(const KArgs p) {
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int PA = A_BYTES / 1024 / 4, PB = B_BYTES / 1024 / 4;          // DMA pieces per wave and stage
    extern __shared__ __attribute__((aligned(16))) char big_lds[];           // 2 stages

    const int tiles_n = (p.N + BN - 1) / BN;
    int tm, z, tn;
    dlsg::gemm_tile_map(tiles_n, tm, z, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    if (p.skip_if && *p.skip_if) return;     // block-uniform: the whole launch is a no-op on this replay
    const int gi = z % p.ngroups, bi = z / p.ngroups;
    const dlsg_gemm_group grp = p.g[gi];
    const float* A = grp.A + (int64_t)bi * p.bsa;
    const float* B = grp.B + (int64_t)bi * p.bsb;
    float* C = grp.C + (int64_t)bi * p.bsc;
    const int K = grp.K;
    const int Ng = grp.N > 0 ? grp.N : p.N;      // this group's output width
    if (n0 >= Ng) return;

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = w >> 1, wn = w & 1;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nst = (K + BK - 1) / BK;
    // pieces [lo, hi) of this wave's share of stage s (A pieces first, then B)
    auto issue = [&](int s, int lo, int hi) {
        const int k0 = s * BK;
        const int koff = (k0 + BK <= K) ? k0 : (K - BK);      // a partial last stage is fetched from K - 32 and masked after the read
        char* st = big_lds + (s & 1) * STAGE;
#pragma unroll
        for (int i = 0; i < PA + PB; ++i) {
            if (i < lo || i >= hi) continue;
            if (i < PA) issue_piece<BM, AT>(A, grp.lda, m0, p.M, koff, w * PA + i, lane, st);
            else issue_piece<BN, BT>(B, grp.ldb, n0, Ng, koff, w * PB + (i - PA), lane, st + A_BYTES);
        }
    };

    // fragment offsets (bytes) of k-group q inside an operand image
    int offA[4], offB[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ks = (2 * q + h) ^ ((r >> 1) & 7);
        offA[q] = AT ? ((4 * (2 * q + h)) * BM + wm * WM + r) * 4 : (wm * WM + r) * 128 + ks * 16;
        offB[q] = BT ? ((4 * (2 * q + h)) * BN + wn * WN + r) * 4 : (wn * WN + r) * 128 + ks * 16;
    }
    f32x4 fa[2][TM], fb[2][TN];
    auto read_q = [&](int buf, const char* st, int q) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (!AT) fa[buf][i] = *reinterpret_cast<const f32x4*>(st + offA[q] + i * 32 * 128);
            else {
#pragma unroll
                for (int j = 0; j < 4; ++j) fa[buf][i][j] = *reinterpret_cast<const float*>(st + offA[q] + (j * BM + i * 32) * 4);
            }
        }
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            if (!BT) fb[buf][i] = *reinterpret_cast<const f32x4*>(st + A_BYTES + offB[q] + i * 32 * 128);
            else {
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[buf][i][j] = *reinterpret_cast<const float*>(st + A_BYTES + offB[q] + (j * BN + i * 32) * 4);
            }
        }
    };
    auto mask_q = [&](int buf, int s, int q) {               // partial last stage: keep k >= k0 only (A side; B is finite)
        const int k0 = s * BK, kbase = K - BK;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool keep = kbase + 4 * (2 * q + h) + j >= k0;
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[buf][i][j] = keep ? fa[buf][i][j] : 0.f;
#pragma unroll
            for (int i = 0; i < TN; ++i) fb[buf][i][j] = keep ? fb[buf][i][j] : 0.f;
        }
    };
    auto mfma_q = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn)
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][i][j], fb[buf][jn][j], acc[i][jn], 0, 0, 0);
    };

    constexpr int PT = PA + PB, PQ = (PT + 3) / 4;             // DMA pieces issued per k-group
    if (nst > 0) issue(0, 0, PT);
    for (int s = 0; s < nst; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of stage s have landed
        __syncthreads();                                       // ... everybody's; and stage s - 1 has been read out
        const char* st = big_lds + (s & 1) * STAGE;
        const bool more = s + 1 < nst;                         // block-uniform
        const bool part = (s + 1) * BK > K;
        read_q(0, st, 0);
        if (part) mask_q(0, s, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q < 3) {
                read_q((q + 1) & 1, st, q + 1);
                if (part) mask_q((q + 1) & 1, s, q + 1);
            }
            if (more) issue(s + 1, q * PQ, min(PT, (q + 1) * PQ));
            mfma_q(q & 1);
        }
    }

    // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
    const float* biasp = grp.bias ? grp.bias : p.bias;
    const bool accum = p.flags & DLSG_GEMM_ACCUM, use_bias = (p.flags & DLSG_GEMM_BIAS) && biasp != nullptr;
    const bool do_tanh = p.flags & DLSG_GEMM_TANH;
    const int64_t ldc = grp.ldc ? grp.ldc : (int64_t)p.ldc;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * WN + j * 32 + r;
            if (col >= Ng) continue;
            const float bv = use_bias ? biasp[col] : 0.f;
            float* cp = C + (int64_t)(m0 + wm * WM + i * 32 + 4 * h) * ldc + col;
            const int rbase = m0 + wm * WM + i * 32 + 4 * h;
            // C += : the 16 old values of the tile are loaded together BEFORE the first store (the compiler cannot prove that a
            // store does not alias the next load and would otherwise wait for 16 dependent round trips per tile)
            float oldv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ro = (e & 3) + 8 * (e >> 2);
                oldv[e] = (accum && rbase + ro < p.M) ? cp[(int64_t)ro * ldc] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ro = (e & 3) + 8 * (e >> 2);
                if (rbase + ro >= p.M) continue;
                float v = p.alpha * acc[i][j][e] + bv + oldv[e];
                if (do_tanh) v = tanhf(v);
                cp[(int64_t)ro * ldc] = v;
            }
        }
}

template <int BM, int BN, bool AT, bool BT>
int launch_one(const KArgs& k, dim3 grid, hipStream_t st) {
    constexpr int lds_bytes = 2 * (BM + BN) * 128;
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_big_kernel<BM, BN, AT, BT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  lds_bytes);
    });
    hipLaunchKernelGGL((gemm_big_kernel<BM, BN, AT, BT>), grid, dim3(NT, 1, 1), lds_bytes, st, k);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}

template <int BM, int BN>
int launch_big(const dlsg_gemm_args* a, hipStream_t st) {
    KArgs k;
    k.M = a->M; k.N = a->N; k.ldc = a->ldc; k.ngroups = a->ngroups; k.flags = a->flags;
    k.bsa = a->bsa; k.bsb = a->bsb; k.bsc = a->bsc; k.alpha = a->alpha; k.bias = a->bias; k.skip_if = a->skip_if;
    for (int i = 0; i < a->ngroups; ++i) k.g[i] = a->g[i];
    const int tiles = ((a->M + BM - 1) / BM) * ((a->N + BN - 1) / BN);
    const dim3 grid(tiles, a->ngroups * a->nbatch, 1);
    switch (a->mode) {
        case 0: return launch_one<BM, BN, false, false>(k, grid, st);
        case 1: return launch_one<BM, BN, false, true>(k, grid, st);
        case 2: return launch_one<BM, BN, true, true>(k, grid, st);
        default: return DLSG_EINVAL;
    }
}

}  // namespace

// 1: the operands of this call meet the large-tile kernels' alignment / size conditions
int dlsg_gemm_big_ok(const dlsg_gemm_args* a) {
    const bool at = a->mode == 2, bt = a->mode != 0;
    if (a->M < 4 || a->N < 4) return 0;
    if (at && (a->M % 4)) return 0;
    if ((a->bsa % 4) || (a->bsb % 4)) return 0;
    for (int i = 0; i < a->ngroups; ++i) {
        const dlsg_gemm_group& g = a->g[i];
        const int gn = g.N > 0 ? g.N : a->N;
        if (g.K < BK || (g.K % 4) || (g.lda % 4) || (g.ldb % 4) || gn < 4) return 0;
        if ((reinterpret_cast<uintptr_t>(g.A) & 15) || (reinterpret_cast<uintptr_t>(g.B) & 15)) return 0;
        if (bt && (gn % 4)) return 0;
    }
    return 1;
}

// bn: 256 or 128 (the tile is 256 x bn)
int dlsg_gemm_big_dispatch(const dlsg_gemm_args* a, hipStream_t st, int bn) {
    if (!dlsg_gemm_big_ok(a)) return DLSG_EINVAL;
    return bn == 256 ? launch_big<256, 256>(a, st) : launch_big<256, 128>(a, st);
}
