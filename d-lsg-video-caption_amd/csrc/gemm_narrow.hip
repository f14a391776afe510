// Narrow products of the critic (DiscV2, models/model.py:110-168, layer.py:661-715) at its three differentiation levels: one side of
// the product is at most 32 wide -- the 26 x 26 word-attention logits, the 26 x 3 word-to-proposal weights, the 512 -> 1 scorers
// and latent-node projections and all their gradient products (tools/critic_gemm_census.py: 78 of the 175 products of an update).
// A 64 x 64 MFMA tile is mostly padding there and the unaligned operands (K = 26, 3, 1) take its scalar path: 15-40 us a launch
// against 5-18 us in rocBLAS.  These are memory / latency bound, not matrix-pipe work (the largest is 0.7 MFLOP per sample): plain
// fp32 FMAs, operands staged through LDS, one pass over the wide operand, coalesced along it.
//
//   kind 1  K <= 32, C = A B (NN) or A^T B (TN):      A (<= 32 x 32 per row block) in LDS, a thread owns two columns of B / C
//   kind 2  N <= 32, C = A B^T (NT), any K:           32 x 32 output tile per workgroup, K in 64-wide LDS stages, 2 x 2 per thread
//   kind 3  M <= 4,  C = A^T B (TN), deep K, no batch: weighted column sums of B; K chunks write partial rows, folded in a fixed order
// Everything else belongs to dlsg_gemm (dlsg_gemm_narrow_kind() == 0).
#include "common.hpp"
#include "dlsg.h"

namespace {

struct NArgs {
    const float* A; const float* B; float* C; const float* bias;
    int64_t lda, ldb, ldc, bsa, bsb, bsc;
    int M, N, K;
    float alpha;
};

// ---------------------------------------------------------------------------------------------- kind 1: K <= 32
template <bool AT>
__global__ __launch_bounds__(256) void narrow_k_kernel(const NArgs p) {
    __shared__ __attribute__((aligned(16))) float At[32][32];           // At[k][m]
    const int b = blockIdx.z, m0 = blockIdx.y * 32, n0 = blockIdx.x * 512;
    const float* A = p.A + b * p.bsa;
    const float* B = p.B + b * p.bsb;
    float* C = p.C + b * p.bsc;
    const int mr = min(32, p.M - m0);
    for (int i = threadIdx.x; i < 32 * 32; i += 256) {
        const int k = i >> 5, m = i & 31;                               // AT: A is (K, M), m contiguous; else (M, K)
        float v = 0.f;
        if (k < p.K && m < mr) v = AT ? A[(int64_t)k * p.lda + m0 + m] : A[(int64_t)(m0 + m) * p.lda + k];
        At[k][m] = v;
    }
    __syncthreads();
    const int c0 = n0 + threadIdx.x, c1 = c0 + 256;
    const bool on0 = c0 < p.N, on1 = c1 < p.N;
    float acc0[32], acc1[32];
#pragma unroll
    for (int m = 0; m < 32; ++m) { acc0[m] = 0.f; acc1[m] = 0.f; }
    const int mq = (mr + 3) >> 2;
    for (int k = 0; k < p.K; ++k) {
        const float b0 = on0 ? B[(int64_t)k * p.ldb + c0] : 0.f;
        const float b1 = on1 ? B[(int64_t)k * p.ldb + c1] : 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (q < mq) {                                               // uniform: rows beyond the block's are never touched
                const f32x4 a = *reinterpret_cast<const f32x4*>(&At[k][4 * q]);
#pragma unroll
                for (int i = 0; i < 4; ++i) { acc0[4 * q + i] += a[i] * b0; acc1[4 * q + i] += a[i] * b1; }
            }
        }
    }
    const float bi0 = (p.bias && on0) ? p.bias[c0] : 0.f, bi1 = (p.bias && on1) ? p.bias[c1] : 0.f;
#pragma unroll
    for (int m = 0; m < 32; ++m) {
        if (m < mr) {
            float* row = C + (int64_t)(m0 + m) * p.ldc;
            if (on0) row[c0] = p.alpha * acc0[m] + bi0;
            if (on1) row[c1] = p.alpha * acc1[m] + bi1;
        }
    }
}

// ---------------------------------------------------------------------------------------------- kind 2: NT, N <= 32
__global__ __launch_bounds__(256) void narrow_n_kernel(const NArgs p) {
    __shared__ float As[32][65];
    __shared__ float Bs[32][65];
    const int b = blockIdx.y, m0 = blockIdx.x * 32;
    const float* A = p.A + b * p.bsa;
    const float* B = p.B + b * p.bsb;
    float* C = p.C + b * p.bsc;
    const int mr = min(32, p.M - m0);
    const int tm = (threadIdx.x >> 4) * 2, tn = (threadIdx.x & 15) * 2;
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    for (int k0 = 0; k0 < p.K; k0 += 64) {
        for (int i = threadIdx.x; i < 32 * 64; i += 256) {
            const int r = i >> 6, k = i & 63;
            const bool kin = k0 + k < p.K;
            As[r][k] = (kin && r < mr) ? A[(int64_t)(m0 + r) * p.lda + k0 + k] : 0.f;
            Bs[r][k] = (kin && r < p.N) ? B[(int64_t)r * p.ldb + k0 + k] : 0.f;
        }
        __syncthreads();
#pragma unroll 16
        for (int k = 0; k < 64; ++k) {
            const float a0 = As[tm][k], a1 = As[tm + 1][k], b0 = Bs[tn][k], b1 = Bs[tn + 1][k];
            acc[0][0] += a0 * b0; acc[0][1] += a0 * b1; acc[1][0] += a1 * b0; acc[1][1] += a1 * b1;
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = tm + i, n = tn + j;
            if (m < mr && n < p.N) C[(int64_t)(m0 + m) * p.ldc + n] = p.alpha * acc[i][j] + (p.bias ? p.bias[n] : 0.f);
        }
}

// ---------------------------------------------------------------------------------------------- kind 3: TN, M <= 4, deep K
constexpr int M3_CHUNK = 64;            // K rows per workgroup
__global__ __launch_bounds__(256) void narrow_m_kernel(const NArgs p, float* part) {
    const int n = blockIdx.x * 256 + threadIdx.x, kc = blockIdx.y;
    const int k0 = kc * M3_CHUNK, k1 = min(p.K, k0 + M3_CHUNK);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (n < p.N) {
        for (int k = k0; k < k1; ++k) {
            const float bv = p.B[(int64_t)k * p.ldb + n];
#pragma unroll
            for (int m = 0; m < 4; ++m)
                if (m < p.M) acc[m] += p.A[(int64_t)k * p.lda + m] * bv;
        }
        float* out = gridDim.y > 1 ? part + (int64_t)kc * p.M * p.N : nullptr;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (m < p.M) {
                if (out) out[m * p.N + n] = acc[m];
                else p.C[(int64_t)m * p.ldc + n] = p.alpha * acc[m];
            }
        }
    }
}
// C[m, n] = alpha * sum over chunks, in chunk order
__global__ void narrow_m_fold_kernel(const float* part, int chunks, NArgs p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.M * p.N) return;
    float s = 0.f;
    for (int c = 0; c < chunks; ++c) s += part[(int64_t)c * p.M * p.N + i];
    p.C[(int64_t)(i / p.N) * p.ldc + (i % p.N)] = p.alpha * s;
}

}  // namespace

extern "C" int dlsg_gemm_narrow_kind(int mode, int M, int N, int K, int nbatch) {
    if (M < 1 || N < 1 || K < 1 || nbatch < 1 || nbatch > 65535) return 0;
    if (mode == 2 && M <= 4 && nbatch == 1 && K > 32) return 3;
    if ((mode == 1 || mode == 2) && K <= 32) return 1;
    if (mode == 0 && N <= 32) return 2;
    return 0;
}
extern "C" int64_t dlsg_gemm_narrow_ws_floats(int mode, int M, int N, int K, int nbatch) {
    if (dlsg_gemm_narrow_kind(mode, M, N, K, nbatch) != 3) return 0;
    const int chunks = (K + M3_CHUNK - 1) / M3_CHUNK;
    return chunks > 1 ? (int64_t)chunks * M * N : 0;
}

extern "C" int dlsg_gemm_narrow(const dlsg_gemm_narrow_args* a, void* stream) {
    if (!a || !a->A || !a->B || !a->C) return DLSG_EINVAL;
    const int kind = dlsg_gemm_narrow_kind(a->mode, a->M, a->N, a->K, a->nbatch);
    if (!kind) return DLSG_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    NArgs p;
    p.A = a->A; p.B = a->B; p.C = a->C; p.bias = a->bias;
    p.lda = a->lda; p.ldb = a->ldb; p.ldc = a->ldc; p.bsa = a->bsa; p.bsb = a->bsb; p.bsc = a->bsc;
    p.M = a->M; p.N = a->N; p.K = a->K; p.alpha = a->alpha;
    if (kind == 1) {
        const dim3 grid((a->N + 511) / 512, (a->M + 31) / 32, a->nbatch);
        if (grid.y > 65535) return DLSG_EINVAL;
        if (a->mode == 2) hipLaunchKernelGGL(narrow_k_kernel<true>, grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL(narrow_k_kernel<false>, grid, dim3(256), 0, st, p);
    } else if (kind == 2) {
        hipLaunchKernelGGL(narrow_n_kernel, dim3((a->M + 31) / 32, a->nbatch), dim3(256), 0, st, p);
    } else {
        if (a->bias) return DLSG_EINVAL;
        const int chunks = (a->K + M3_CHUNK - 1) / M3_CHUNK;
        if (chunks > 1 && (!a->ws || a->ws_floats < (int64_t)chunks * a->M * a->N)) return DLSG_EINVAL;
        hipLaunchKernelGGL(narrow_m_kernel, dim3((a->N + 255) / 256, chunks), dim3(256), 0, st, p, a->ws);
        if (chunks > 1) {
            DLSG_CHECK_LAUNCH();
            hipLaunchKernelGGL(narrow_m_fold_kernel, dim3((a->M * a->N + 255) / 256), dim3(256), 0, st, a->ws, chunks, p);
        }
    }
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
