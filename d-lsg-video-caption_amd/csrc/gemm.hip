// Grouped / batched fp32 GEMM on the gfx950 f32-input matrix cores (v_mfma_f32_32x32x2_f32).
//
// Every dense product on the D-LSG path runs through this one kernel family (see include/dlsg.h):
// projections (NT), input gradients (NN) and weight gradients (TN).  fp32 in / fp32 accumulate keeps the
// reference's fp32 numerics (bitwise a k-ordered fmaf chain), at the fp32 matrix rate (157 TFLOP/s peak).
//
// Layout: 256 threads = 4 waves as 2x2; block tile BMxBN (128x128 or 64x64), BK = 32.
//   * global -> registers -> LDS staging with the next K-tile's loads in flight under the MFMAs
//   * k-contiguous operands are stored [row][32+4] (b128 fragment reads, conflict free: slot stride 9 mod 16)
//     m/n-contiguous operands are stored [k][rows+4] (b32 fragment reads, lanes on consecutive banks)
//   * lane (r = lane&31, h = lane>>5) feeds k = 16h + s for MFMA s of a K-tile: A and B use the same k order
//   * block ids are remapped so that the blocks an XCD receives (id % 8) walk consecutive tiles of one A row
//     panel: the panel stays in that XCD's L2 instead of being fetched by all eight.
#include <cstdlib>
#include <mutex>

#include "common.hpp"
#include "dlsg.h"

namespace {

constexpr int NT = 256;

template <int ROWS, bool T, int BK>
struct TileGeom {
    // float4 count per tile and per thread
    static constexpr int NV = ROWS * BK / 4 / NT;
    static constexpr int LD = T ? (ROWS + 4) : (BK + 4);
    static constexpr int ELEMS = T ? BK * (ROWS + 4) : ROWS * (BK + 4);
    static constexpr int KQ = BK / 4;
};

// Load one operand tile (ROWS x BK) into registers.  T=false: element (row,k) at base[row*ld + k];
// T=true: element (row,k) at base[k*ld + row].
// Fast path (interior tile, 16-B aligned): unconditional float4 loads, no branches -- hipcc otherwise wraps every guarded
// load in its own exec-mask branch and the loads of a tile no longer overlap.  Rows past the matrix edge are clamped to
// the last valid row: they only feed output rows/columns that are never stored.
template <int ROWS, bool T, int BK>
__device__ __forceinline__ void load_tile_fast(const float* __restrict__ base, int64_t ld, int row0, int k0, int rmax,
                                               f32x4 (&regs)[TileGeom<ROWS, T, BK>::NV]) {
    constexpr int NV = TileGeom<ROWS, T, BK>::NV;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int f = threadIdx.x + NT * j;
        const float* ptr;
        if (!T) {
            const int gr = min(row0 + f / (BK / 4), rmax - 1);
            ptr = base + (int64_t)gr * ld + k0 + 4 * (f % (BK / 4));
        } else {
            ptr = base + (int64_t)(k0 + f / (ROWS / 4)) * ld + row0 + 4 * (f % (ROWS / 4));
        }
        regs[j] = *reinterpret_cast<const f32x4*>(ptr);
    }
}
// Slow path: edge tiles / unaligned operands.  Out-of-range elements read as zero.
template <int ROWS, bool T, int BK>
__device__ __forceinline__ void load_tile_slow(const float* __restrict__ base, int64_t ld, int row0, int k0, int rmax,
                                               int K, f32x4 (&regs)[TileGeom<ROWS, T, BK>::NV]) {
    constexpr int NV = TileGeom<ROWS, T, BK>::NV;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int f = threadIdx.x + NT * j;
        int gr, gk, nvalid;
        const float* ptr;
        if (!T) {
            const int row = f / (BK / 4), kq = f % (BK / 4);
            gr = row0 + row;
            gk = k0 + 4 * kq;
            ptr = base + (int64_t)gr * ld + gk;
            nvalid = (gr < rmax) ? min(max(K - gk, 0), 4) : 0;
        } else {
            const int k = f / (ROWS / 4), mq = f % (ROWS / 4);
            gk = k0 + k;
            gr = row0 + 4 * mq;
            ptr = base + (int64_t)gk * ld + gr;
            nvalid = (gk < K) ? min(max(rmax - gr, 0), 4) : 0;
        }
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (nvalid > 0) v[0] = ptr[0];
        if (nvalid > 1) v[1] = ptr[1];
        if (nvalid > 2) v[2] = ptr[2];
        if (nvalid > 3) v[3] = ptr[3];
        regs[j] = v;
    }
}
template <int ROWS, bool T, int BK>
__device__ __forceinline__ void load_tile(const float* __restrict__ base, int64_t ld, int row0, int k0, int rmax,
                                          int K, bool vec_ok, f32x4 (&regs)[TileGeom<ROWS, T, BK>::NV]) {
    // wave-uniform choice per tile
    const bool fast = vec_ok && (k0 + BK <= K) && (T ? (row0 + ROWS <= rmax) : (rmax > 0));
    if (fast) load_tile_fast<ROWS, T, BK>(base, ld, row0, k0, rmax, regs);
    else load_tile_slow<ROWS, T, BK>(base, ld, row0, k0, rmax, K, regs);
}

template <int ROWS, bool T, int BK>
__device__ __forceinline__ void store_tile(float* __restrict__ lds, const f32x4 (&regs)[TileGeom<ROWS, T, BK>::NV]) {
    constexpr int NV = TileGeom<ROWS, T, BK>::NV;
    constexpr int LD = TileGeom<ROWS, T, BK>::LD;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int f = threadIdx.x + NT * j;
        int off;
        if (!T) {
            off = (f / (BK / 4)) * LD + 4 * (f % (BK / 4));
        } else {
            off = (f / (ROWS / 4)) * LD + 4 * (f % (ROWS / 4));
        }
        *reinterpret_cast<f32x4*>(lds + off) = regs[j];
    }
}

// Fragment of 16 k-values for one 32-row subtile: element s <-> k = kbase + 16h + s.
template <int ROWS, bool T, int BK>
__device__ __forceinline__ void read_frag(const float* __restrict__ lds, int rowbase, int kbase, int r, int h, float (&a)[16]) {
    constexpr int LD = TileGeom<ROWS, T, BK>::LD;
    if (!T) {
        const float* p = lds + (rowbase + r) * LD + kbase + 16 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * q);
            a[4 * q + 0] = v[0]; a[4 * q + 1] = v[1]; a[4 * q + 2] = v[2]; a[4 * q + 3] = v[3];
        }
    } else {
        const float* p = lds + (kbase + 16 * h) * LD + rowbase + r;
#pragma unroll
        for (int s = 0; s < 16; ++s) a[s] = p[s * LD];
    }
}

struct KArgs {
    int M, N, ldc, ngroups, flags;
    int64_t bsa, bsb, bsc;
    float alpha;
    const float* bias;
    const int32_t* skip_if;
    dlsg_gemm_group g[DLSG_GEMM_MAXG];
};

template <int BM, int BN, bool AT, bool BT, int BK>
__device__ __forceinline__ void gemm_body(const KArgs& p) {
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int TM = WM / 32, TN = WN / 32;
    using GA = TileGeom<BM, AT, BK>;
    using GB = TileGeom<BN, BT, BK>;
    __shared__ __attribute__((aligned(16))) float lds[GA::ELEMS + GB::ELEMS];
    float* ldsA = lds;
    float* ldsB = lds + GA::ELEMS;

    // ---- XCD-aware tile order (common.hpp)
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
    const bool vecA = ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && ((grp.lda & 3) == 0);
    const bool vecB = ((reinterpret_cast<uintptr_t>(B) & 15) == 0) && ((grp.ldb & 3) == 0);

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

    f32x4 ra[GA::NV], rb[GB::NV];
    const int nk = (K + BK - 1) / BK;
    if (nk > 0) {
        load_tile<BM, AT, BK>(A, grp.lda, m0, 0, p.M, K, vecA, ra);
        load_tile<BN, BT, BK>(B, grp.ldb, n0, 0, Ng, K, vecB, rb);
    }
    for (int kt = 0; kt < nk; ++kt) {
        store_tile<BM, AT, BK>(ldsA, ra);
        store_tile<BN, BT, BK>(ldsB, rb);
        __syncthreads();
        if (kt + 1 < nk) {
            load_tile<BM, AT, BK>(A, grp.lda, m0, (kt + 1) * BK, p.M, K, vecA, ra);
            load_tile<BN, BT, BK>(B, grp.ldb, n0, (kt + 1) * BK, Ng, K, vecB, rb);
        }
#pragma unroll
        for (int sub = 0; sub < BK / 32; ++sub) {      // 32-deep slices of the K tile
            float fa[TM][16], fb[TN][16];
#pragma unroll
            for (int i = 0; i < TM; ++i) read_frag<BM, AT, BK>(ldsA, wm * WM + i * 32, 32 * sub, r, h, fa[i]);
#pragma unroll
            for (int j = 0; j < TN; ++j) read_frag<BN, BT, BK>(ldsB, wn * WN + j * 32, 32 * sub, r, h, fb[j]);
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
    const float* biasp = grp.bias ? grp.bias : p.bias;
    const bool accum = p.flags & DLSG_GEMM_ACCUM, use_bias = (p.flags & DLSG_GEMM_BIAS) && biasp != nullptr;
    const bool do_tanh = p.flags & DLSG_GEMM_TANH;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * WN + j * 32 + r;
            if (col >= Ng) continue;
            const float bv = use_bias ? biasp[col] : 0.f;
            const int64_t ldc = grp.ldc ? grp.ldc : (int64_t)p.ldc;
            const int rbase = m0 + wm * WM + i * 32 + 4 * h;
            float* cp = C + (int64_t)rbase * ldc + col;
            // C += : the tile's 16 old values are loaded together BEFORE the first store (a store may alias the next load as far
            // as the compiler knows: 16 dependent round trips per tile otherwise)
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

// Two entry points over one body: the 128 x 128 tile is compiled for 3 waves per SIMD (<= 170 registers; the compiler's
// own choice is 182-196, i.e. two workgroups per CU): measured +7 % on the region projection and +11 % on deep TN
// products (tools/archive/gemm_fp32_probe.py); the 64 x 64 tile already fits 3-4 waves and keeps the compiler's allocation.
template <int BM, int BN, bool AT, bool BT, int BK>
__global__ __launch_bounds__(NT) void gemm_kernel(const KArgs p) { gemm_body<BM, BN, AT, BT, BK>(p); }
template <int BM, int BN, bool AT, bool BT, int BK>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(3, 3))) void gemm_kernel_w3(const KArgs p) { gemm_body<BM, BN, AT, BT, BK>(p); }

// ------------------------------------------------------------------------------------------------ skinny GEMM (M <= 128)
// The recurrent products of the path (BiLSTM / LSTMCell gates and their input gradients) have M = batch rows: every weight
// element is used once per launch, so they are weight-streaming and latency-bound, not MFMA-bound.
// Block = 4 waves on one (32 MI) x 32 output tile (MI = 2: batches up to 64 rows; MI = 4: up to 128 -- the weight stream of a
// launch is the same, twice the rows ride on it), walking K in 128-wide super-chunks:
//   * the activation chunk (64 x 128) and, for k-contiguous weights (NT), the weight chunk (32 x 128) are staged through
//     LDS with fully coalesced loads (32 lanes x 16 B = one 512-B row segment); a lane-per-row gather straight into MFMA
//     fragments was measured first: 64 cache lines per load instruction, address-coalescer bound (34 us vs 20 us);
//   * n-contiguous weights (NN) are read straight into registers: lanes already run along n;
//   * the 4 waves split the super-chunk's K (wave w owns k in [32w, 32w+32)), so a wave issues 32 fp32 MFMAs (or 12
//     bf16 MFMAs on the split-bf16 path) per super-chunk; the next super-chunk's loads are in flight meanwhile;
//   * the 4 partial tiles are summed through LDS in a fixed order (deterministic).
constexpr int SK = 128;            // super-chunk depth
constexpr int SLD = SK + 4;        // LDS row stride (floats): 16-B slot stride 33 = 1 mod 16

constexpr int skinny_lds_bytes(int MI) { return (32 * MI + 32) * SLD * 4; }      // 50,688 B (MI = 2) / 84,480 B (MI = 4)

template <bool BT, int MI>
__global__ __launch_bounds__(NT) void skinny_kernel(const KArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];             // (32 MI + 32) x SLD floats; reused for the final reduction
    float* ldsA = lds;
    float* ldsB = lds + 32 * MI * SLD;
    if (p.skip_if && *p.skip_if) return;     // block-uniform: the whole launch is a no-op on this replay
    const int z = blockIdx.y;
    const int gi = z % p.ngroups, bi = z / p.ngroups;
    const dlsg_gemm_group grp = p.g[gi];
    const float* A = grp.A + (int64_t)bi * p.bsa;
    const float* B = grp.B + (int64_t)bi * p.bsb;
    float* C = grp.C + (int64_t)bi * p.bsc;
    const int K = grp.K, M = p.M, N = grp.N > 0 ? grp.N : p.N;
    const int n0 = blockIdx.x * 32;
    if (n0 >= N) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const bool vecA = ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && ((grp.lda & 3) == 0);
    const bool vecB = ((reinterpret_cast<uintptr_t>(B) & 15) == 0) && ((grp.ldb & 3) == 0);
    const int col = n0 + r;

    f32x16 acc[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    f32x4 ra[4 * MI], rb[4];     // staged operands of the next super-chunk
    float bdir[16];              // NN: this lane's 16 weight values of the next super-chunk

    auto ld4s = [&](const float* ptr, int nvalid) {      // guarded (edge) load
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (nvalid > 0) v[0] = ptr[0];
        if (nvalid > 1) v[1] = ptr[1];
        if (nvalid > 2) v[2] = ptr[2];
        if (nvalid > 3) v[3] = ptr[3];
        return v;
    };
    auto load_super = [&](int k0) {
        const bool fullk = k0 + SK <= K;                  // wave-uniform: branch-free loads on interior super-chunks
        if (fullk && vecA) {
#pragma unroll
            for (int j = 0; j < 4 * MI; ++j) {
                const int f = threadIdx.x + NT * j;
                const int row = min(f >> 5, M - 1);       // rows >= M feed output rows that are never stored
                ra[j] = *reinterpret_cast<const f32x4*>(A + (int64_t)row * grp.lda + k0 + 4 * (f & 31));
            }
        } else if (vecA && (K & 3) == 0 && K >= 4) {       // aligned tail: whole float4s, clamped address + select
#pragma unroll
            for (int j = 0; j < 4 * MI; ++j) {
                const int f = threadIdx.x + NT * j;
                const int row = min(f >> 5, M - 1), k = k0 + 4 * (f & 31);
                const f32x4 v = *reinterpret_cast<const f32x4*>(A + (int64_t)row * grp.lda + min(k, K - 4));
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                ra[j] = (k < K) ? v : zero;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4 * MI; ++j) {
                const int f = threadIdx.x + NT * j;
                const int row = f >> 5, k = k0 + 4 * (f & 31);
                ra[j] = ld4s(A + (int64_t)row * grp.lda + k, row < M ? min(max(K - k, 0), 4) : 0);
            }
        }
        if (!BT) {
            if (fullk && vecB) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int f = threadIdx.x + NT * j;
                    const int row = min(n0 + (f >> 5), N - 1);
                    rb[j] = *reinterpret_cast<const f32x4*>(B + (int64_t)row * grp.ldb + k0 + 4 * (f & 31));
                }
            } else if (vecB && (K & 3) == 0 && K >= 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int f = threadIdx.x + NT * j;
                    const int row = min(n0 + (f >> 5), N - 1), k = k0 + 4 * (f & 31);
                    const f32x4 v = *reinterpret_cast<const f32x4*>(B + (int64_t)row * grp.ldb + min(k, K - 4));
                    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                    rb[j] = (k < K) ? v : zero;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int f = threadIdx.x + NT * j;
                    const int row = f >> 5, k = k0 + 4 * (f & 31);
                    rb[j] = ld4s(B + (int64_t)(n0 + row) * grp.ldb + k, (n0 + row) < N ? min(max(K - k, 0), 4) : 0);
                }
            }
        } else {
            const int cc = min(col, N - 1);
            if (fullk) {
#pragma unroll
                for (int s2 = 0; s2 < 16; ++s2) bdir[s2] = B[(int64_t)(k0 + 32 * w + 16 * h + s2) * grp.ldb + cc];
            } else {
#pragma unroll
                for (int s2 = 0; s2 < 16; ++s2) {
                    const int k = k0 + 32 * w + 16 * h + s2;
                    const float v = B[(int64_t)min(k, K - 1) * grp.ldb + cc];      // clamped address + select: no branch
                    bdir[s2] = (k < K) ? v : 0.f;
                }
            }
        }
    };
    auto store_super = [&]() {
#pragma unroll
        for (int j = 0; j < 4 * MI; ++j) {
            const int f = threadIdx.x + NT * j;
            *reinterpret_cast<f32x4*>(ldsA + (f >> 5) * SLD + 4 * (f & 31)) = ra[j];
        }
        if (!BT) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f = threadIdx.x + NT * j;
                *reinterpret_cast<f32x4*>(ldsB + (f >> 5) * SLD + 4 * (f & 31)) = rb[j];
            }
        }
    };

    const int nsup = (K + SK - 1) / SK;
    if (nsup > 0) load_super(0);
    for (int sc = 0; sc < nsup; ++sc) {
        store_super();
        float bcur[16];
        if (BT) {
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) bcur[s2] = bdir[s2];
        }
        __syncthreads();
        if (sc + 1 < nsup) load_super((sc + 1) * SK);
        // fragments of this wave's 32-deep slice: lane half h owns k = 32w + 16h + s
        float fa[MI][16];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const float* ap = ldsA + (r + 32 * mi) * SLD + 32 * w + 16 * h;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(ap + 4 * q);
                fa[mi][4 * q] = v[0]; fa[mi][4 * q + 1] = v[1]; fa[mi][4 * q + 2] = v[2]; fa[mi][4 * q + 3] = v[3];
            }
        }
        if (!BT) {
            const float* bp = ldsB + r * SLD + 32 * w + 16 * h;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(bp + 4 * q);
                bcur[4 * q] = v[0]; bcur[4 * q + 1] = v[1]; bcur[4 * q + 2] = v[2]; bcur[4 * q + 3] = v[3];
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi][s2], bcur[s2], acc[mi], 0, 0, 0);
        __syncthreads();
    }
    // ---- sum the 4 waves' partial tiles through LDS (staging buffers are free now); wave w finalises e in [4w, 4w+4)
    float* red = lds;               // [4][MI][16][64] floats = MI x 16 KB
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int e = 0; e < 16; ++e) red[((w * MI + mi) * 16 + e) * 64 + lane] = acc[mi][e];
    __syncthreads();
    const float* biasp = grp.bias ? grp.bias : p.bias;
    const bool accum = p.flags & DLSG_GEMM_ACCUM, use_bias = (p.flags & DLSG_GEMM_BIAS) && biasp != nullptr;
    const bool do_tanh = p.flags & DLSG_GEMM_TANH;
    if (col < N) {
        const float bv = use_bias ? biasp[col] : 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ee = 0; ee < 4; ++ee) {
                const int e = 4 * w + ee;
                const int row = 32 * mi + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row >= M) continue;
                float v = 0.f;
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) v += red[((ww * MI + mi) * 16 + e) * 64 + lane];
                v = p.alpha * v + bv;
                float* cp = C + (int64_t)row * (grp.ldc ? grp.ldc : (int64_t)p.ldc) + col;
                if (accum) v += *cp;
                if (do_tanh) v = tanhf(v);
                *cp = v;
            }
    }
}

// ------------------------------------------------------------------------------------------------ skinny GEMM, NT, generation 2
// Same decomposition as skinny_kernel<false> ((32 MI) x 32 output tile per workgroup, the 4 waves split K, K-split groups from the
// caller), different pipeline: the first kernel staged global -> registers -> LDS with two barriers per 128-deep super-chunk
// and one chunk of prefetch; at two workgroups per CU that left the matrix pipe ~40 % busy on the recurrent products
// (23 us per launch against a ~11 us pipe floor).  Here every wave streams ITS OWN k-slices and nobody else's:
//   * stage = 16 k of one wave: A (64 rows x 64 B) + B (32 rows x 64 B) = 6 LDS-DMA pieces (`global_load_lds_dwordx4`, 16 rows x
//     4 x 16 B each: 64-B row segments, half a cache line, the other half is the wave's next stage), no staging registers;
//   * a private 3-stage ring per wave (18 KB; 72 KB per workgroup, two workgroups per CU): two stages in flight under the
//     16 MFMAs of the current one, ordered by the wave's own counted vmcnt -- NO barrier in the K loop;
//   * the LDS image is lane-linear per piece (the DMA cannot scatter), so the bank swizzle is applied on the SOURCE address:
//     slot (row, s) of a piece holds k-segment s ^ ((row >> 2) & 3); fragment reads (b128, lane = row) are conflict-free;
//   * k order inside a stage: lane half h supplies k-segments 2q + h, the same for A and B.
// Needs 16-B aligned, 4-float-strided operands and K % 4 == 0 (a partial last stage is masked after the read); anything
// else goes to skinny_kernel.
// Measured (tools/archive/recurrent_gemm_bench.py, batch 64): query gates 21.5 us (first kernel 23.5), BiLSTM step 16.8 (18.8), language
// gates 30.4 (29.8).  What bounds it is NOT the pipeline: with every DMA address pinned to cache-hot lines AND the MFMAs
// removed, the loop still takes 20 of the 30 us -- the per-CU global -> LDS path (~40 GB/s per CU in this access shape) has to
// move 6 KB per 16 MFMAs at M = 64, two thirds of it the activations that every 32-column tile re-fetches.  An
// activation-stationary variant (the wave's 64 x K/4 activation slices in 128 registers, several column tiles per workgroup,
// weights only through the ring) was built and dropped: 37-46 us -- all workgroups sit in their activation prologue at the
// same time, then all in their MFMA phase, and the per-tile 4-way partial sums add barriers that one wave per SIMD cannot hide.
constexpr int S2_STAGE = 16;                 // k per wave and stage
constexpr int S2_DEPTH = 3;                  // ring slots per wave
constexpr int S2_B_BYTES = 32 * 64;          // one 32-row block of a stage (A or B): 32 rows x 64 B
constexpr int s2_lds_bytes(int MI, int NJ) { return 4 * S2_DEPTH * (MI + NJ) * S2_B_BYTES; }
// MI = 2: 73,728 B (NJ = 1) / 98,304 B (NJ = 2);  MI = 4 (batches of 65..128 rows): 122,880 B / 147,456 B

template <int VB>
__device__ __forceinline__ void s2_glds(const char* src, char* dst) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) void* lp_t;
    typedef const __attribute__((address_space(1))) void* gp_t;
    __builtin_amdgcn_global_load_lds((gp_t)src, (lp_t)dst, 16, 0, 0);
#endif
}

// NJ = 32-column blocks per workgroup: NJ = 2 halves the activation bytes per flop (the A stage feeds two column blocks);
// used when the launch still fills the chip with half as many workgroups.  (NJ = 4 -- 64 x 128 tiles, 144 KB of rings, the
// contraction in 8 groups of 512 to keep 256 workgroups -- was measured and dropped: language gates 32.5 us against 28.6, query
// gates 29.4 against 23.8: eight 16-deep stages per wave no longer amortise the ring's ramp.  A fourth ring slot (three stages
// in flight, 128 KB) changed nothing either: 21.6 / 28.6 / 16.2 us against 21.6 / 28.0 / 17.3; nor did giving each wave a
// contiguous k range, so that its consecutive stages are the two halves of the same 128-B lines: 22.3 / 28.9 / 16.4; nor did
// issuing the next stage's pieces one by one between the MFMA groups instead of together: 22.0 / 29.6 / 17.4.  The launch is
// as long as its bytes take through the CUs' load paths, whatever the schedule around them.)
// BT: the weights are (K x N) row-major (the NN form: input gradients): a stage's weight block is [16 k][32 columns], one 128-B
// row per k, read by b32 with lanes on consecutive banks; the activation side is the same.  (Until round 4 the NN form ran on the
// register-staged first kernel: 32.1 us against 26.5 for the same 64 x 4096 x 4096 product.)
template <int MI, int NJ, bool BT = false>
__global__ __launch_bounds__(NT) void skinny2_nt_kernel(const KArgs p) {
    constexpr int S2_A_BYTES = MI * S2_B_BYTES;
    constexpr int SLOT = S2_A_BYTES + NJ * S2_B_BYTES;
    extern __shared__ __attribute__((aligned(16))) char s2_lds[];
    if (p.skip_if && *p.skip_if) return;     // block-uniform: the whole launch is a no-op on this replay
    const int z = blockIdx.y;
    const int gi = z % p.ngroups, bi = z / p.ngroups;
    const dlsg_gemm_group grp = p.g[gi];
    const float* A = grp.A + (int64_t)bi * p.bsa;
    const float* B = grp.B + (int64_t)bi * p.bsb;
    float* C = grp.C + (int64_t)bi * p.bsc;
    const int K = grp.K, M = p.M, N = grp.N > 0 ? grp.N : p.N;
    const int n0 = blockIdx.x * 32 * NJ;
    if (n0 >= N) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    char* ring = s2_lds + w * (S2_DEPTH * SLOT);

    // ---- per-lane source rows of the pieces of a stage (row and swizzle never change; only k0 does)
    const int rho = lane >> 2, sig = lane & 3;
    const float* srcA[2 * MI];
    const float* srcB[2 * NJ];
#pragma unroll
    for (int pc = 0; pc < 2 * MI; ++pc) {
        const int R = 16 * pc + rho;
        srcA[pc] = A + (int64_t)min(R, M - 1) * grp.lda + 4 * (sig ^ ((R >> 2) & 3));
    }
#pragma unroll
    for (int pc = 0; pc < 2 * NJ; ++pc) {
        if (!BT) {
            const int R = 16 * pc + rho;
            srcB[pc] = B + (int64_t)min(n0 + R, N - 1) * grp.ldb + 4 * (sig ^ ((R >> 2) & 3));
        } else {
            // piece pc = k-rows 8 (pc & 1) .. + 7 of column block pc >> 1; lane -> (k-row, 16-B segment = 4 columns)
            const int kr = 8 * (pc & 1) + (lane >> 3), c = n0 + 32 * (pc >> 1) + 4 * (lane & 7);
            srcB[pc] = B + (int64_t)kr * grp.ldb + min(c, N - 4);
        }
    }
    const int nst = (K + S2_STAGE - 1) / S2_STAGE;            // stages of the whole K range
    const int mine = (nst - w + 3) / 4;                        // stages s = w, w + 4, ... of this wave
    auto issue = [&](int i) {                                  // i-th stage of this wave -> ring slot i % DEPTH
        const int k0 = (4 * i + w) * S2_STAGE;
        char* slot = ring + (i % S2_DEPTH) * SLOT;
        // a partial last stage is fetched from K - 16 (every 16-B read stays inside the row) and masked after the read
        const int koff = (k0 + S2_STAGE <= K) ? k0 : (K - S2_STAGE);
#pragma unroll
        for (int pc = 0; pc < 2 * MI; ++pc) s2_glds<16>(reinterpret_cast<const char*>(srcA[pc] + koff), slot + pc * 1024);
#pragma unroll
        for (int pc = 0; pc < 2 * NJ; ++pc)
            s2_glds<16>(reinterpret_cast<const char*>(BT ? srcB[pc] + (int64_t)koff * grp.ldb : srcB[pc] + koff), slot + S2_A_BYTES + pc * 1024);
    };

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][nj][e] = 0.f;

    // fragment addresses inside a slot (bytes): row R, k-segment ks -> (R >> 4) * 1024 + (R & 15) * 64 + (ks ^ ((R >> 2) & 3)) * 16
    int offA[MI][2], offB[NJ][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int ks = 2 * q + h;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int R = 32 * mi + r;
            offA[mi][q] = (R >> 4) * 1024 + (R & 15) * 64 + ((ks ^ ((R >> 2) & 3)) * 16);
        }
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) {
            const int R = 32 * nj + r;
            offB[nj][q] = BT ? S2_A_BYTES + nj * 2048 + (4 * ks) * 128 + r * 4
                             : S2_A_BYTES + (R >> 4) * 1024 + (R & 15) * 64 + ((ks ^ ((R >> 2) & 3)) * 16);
        }
    }
    constexpr int PCS = 2 * MI + 2 * NJ;                         // DMA pieces per stage

    if (mine > 0) issue(0);
    if (mine > 1) issue(1);
    for (int i = 0; i < mine; ++i) {
        if (i + 2 < mine) {
            issue(i + 2);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PCS) : "memory");      // stage i landed, two stages stay in flight
        } else if (i + 1 < mine) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PCS) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const char* slot = ring + (i % S2_DEPTH) * SLOT;
        const int k0 = (4 * i + w) * S2_STAGE;
        const bool part = k0 + S2_STAGE > K;                     // wave-uniform
        const int kbase = part ? K - S2_STAGE : k0;              // where the DMA really read
        f32x4 fa[MI][2], fb[NJ][2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) fa[mi][q] = *reinterpret_cast<const f32x4*>(slot + offA[mi][q]);
#pragma unroll
            for (int nj = 0; nj < NJ; ++nj) {
                if (!BT) fb[nj][q] = *reinterpret_cast<const f32x4*>(slot + offB[nj][q]);
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb[nj][q][j] = *reinterpret_cast<const float*>(slot + offB[nj][q] + j * 128);
                }
            }
        }
        if (part) {
            // the stage was fetched from kbase (< k0): keep only k in [k0, K), zero the rest (already counted by earlier stages)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = kbase + 4 * (2 * q + h) + j;
                    const bool keep = k >= k0 && k < K;
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) fa[mi][q][j] = keep ? fa[mi][q][j] : 0.f;
#pragma unroll
                    for (int nj = 0; nj < NJ; ++nj) fb[nj][q][j] = keep ? fb[nj][q][j] : 0.f;
                }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        acc[mi][nj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi][q][j], fb[nj][q][j], acc[mi][nj], 0, 0, 0);
    }
    // ---- sum the 4 waves' partial tiles through LDS (the rings are free now); wave w finalises e in [4w, 4w+4)
    __syncthreads();
    float* red = reinterpret_cast<float*>(s2_lds);               // [4][MI][NJ][16][64] floats = MI x NJ x 16 KB
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
            for (int e = 0; e < 16; ++e) red[(((w * MI + mi) * NJ + nj) * 16 + e) * 64 + lane] = acc[mi][nj][e];
    __syncthreads();
    const float* biasp = grp.bias ? grp.bias : p.bias;
    const bool accum = p.flags & DLSG_GEMM_ACCUM, use_bias = (p.flags & DLSG_GEMM_BIAS) && biasp != nullptr;
    const bool do_tanh = p.flags & DLSG_GEMM_TANH;
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) {
        const int col = n0 + 32 * nj + r;
        if (col >= N) continue;
        const float bv = use_bias ? biasp[col] : 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ee = 0; ee < 4; ++ee) {
                const int e = 4 * w + ee;
                const int row = 32 * mi + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row >= M) continue;
                float v = 0.f;
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) v += red[(((ww * MI + mi) * NJ + nj) * 16 + e) * 64 + lane];
                v = p.alpha * v + bv;
                float* cp = C + (int64_t)row * (grp.ldc ? grp.ldc : (int64_t)p.ldc) + col;
                if (accum) v += *cp;
                if (do_tanh) v = tanhf(v);
                *cp = v;
            }
    }
}

template <int MI>
int launch_skinny_mi(const dlsg_gemm_args* a, const KArgs& k, hipStream_t st) {
    dim3 grid((a->N + 31) / 32, a->ngroups * a->nbatch, 1), block(NT, 1, 1);
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny2_nt_kernel<MI, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  s2_lds_bytes(MI, 1));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny2_nt_kernel<MI, 2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  s2_lds_bytes(MI, 2));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny2_nt_kernel<MI, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  s2_lds_bytes(MI, 1));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny2_nt_kernel<MI, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  s2_lds_bytes(MI, 2));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny_kernel<false, MI>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  skinny_lds_bytes(MI));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny_kernel<true, MI>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  skinny_lds_bytes(MI));
    });
    if (a->mode == 0) {
        // generation 2 (LDS-DMA rings, no barrier in the K loop) when every operand is 16-B aligned with 4-float strides
        bool ok = a->nbatch == 1;
        for (int i = 0; i < a->ngroups && ok; ++i) {
            const dlsg_gemm_group& g = a->g[i];
            ok = (g.K % 4 == 0) && g.K >= S2_STAGE && (g.lda % 4 == 0) && (g.ldb % 4 == 0) &&
                 ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0) && ((reinterpret_cast<uintptr_t>(g.B) & 15) == 0);
        }
        if (ok) {
            // 64-column workgroups (one per CU) when they still put >= ~one workgroup on every CU
            const int wg64 = ((a->N + 63) / 64) * a->ngroups * a->nbatch;
            if (wg64 >= 224) {
                dim3 grid2((a->N + 63) / 64, a->ngroups * a->nbatch, 1);
                hipLaunchKernelGGL((skinny2_nt_kernel<MI, 2>), grid2, block, s2_lds_bytes(MI, 2), st, k);
            } else {
                hipLaunchKernelGGL((skinny2_nt_kernel<MI, 1>), grid, block, s2_lds_bytes(MI, 1), st, k);
            }
        } else {
            hipLaunchKernelGGL((skinny_kernel<false, MI>), grid, block, skinny_lds_bytes(MI), st, k);
        }
    } else {
        // NN form on the same LDS-DMA ring when the operands are 16-B aligned and the widths multiples of 4
        bool ok = a->nbatch == 1;
        for (int i = 0; i < a->ngroups && ok; ++i) {
            const dlsg_gemm_group& g = a->g[i];
            const int gn = g.N > 0 ? g.N : a->N;
            ok = (g.K % 4 == 0) && g.K >= S2_STAGE && (g.lda % 4 == 0) && (g.ldb % 4 == 0) && (gn % 4 == 0) && gn >= 4 &&
                 ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0) && ((reinterpret_cast<uintptr_t>(g.B) & 15) == 0);
        }
        if (ok) {
            // 64-column workgroups (one per CU) only when every group is as wide as the launch: groups of different widths leave
            // empty workgroups in the grid, and at one workgroup per CU the real ones behind them wait for a second round
            // (the language cell's [W_ih | W_hh] gradients: 49 us on 64-column workgroups, 32 on the first kernel)
            bool same_n = true;
            for (int i = 0; i < a->ngroups; ++i) same_n = same_n && (a->g[i].N == 0 || a->g[i].N == a->N);
            const int wg64 = ((a->N + 63) / 64) * a->ngroups * a->nbatch;
            if (wg64 >= 224 && same_n) {
                dim3 grid2((a->N + 63) / 64, a->ngroups * a->nbatch, 1);
                hipLaunchKernelGGL((skinny2_nt_kernel<MI, 2, true>), grid2, block, s2_lds_bytes(MI, 2), st, k);
            } else {
                hipLaunchKernelGGL((skinny2_nt_kernel<MI, 1, true>), grid, block, s2_lds_bytes(MI, 1), st, k);
            }
        } else {
            hipLaunchKernelGGL((skinny_kernel<true, MI>), grid, block, skinny_lds_bytes(MI), st, k);
        }
    }
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}

int launch_skinny(const dlsg_gemm_args* a, hipStream_t st) {
    KArgs k;
    k.M = a->M; k.N = a->N; k.ldc = a->ldc; k.ngroups = a->ngroups; k.flags = a->flags;
    k.bsa = a->bsa; k.bsb = a->bsb; k.bsc = a->bsc; k.alpha = a->alpha; k.bias = a->bias; k.skip_if = a->skip_if;
    for (int i = 0; i < a->ngroups; ++i) k.g[i] = a->g[i];
    return a->M <= 64 ? launch_skinny_mi<2>(a, k, st) : launch_skinny_mi<4>(a, k, st);
}

template <int BM, int BN, int BK>
int launch(const dlsg_gemm_args* a, hipStream_t st) {
    KArgs k;
    k.M = a->M; k.N = a->N; k.ldc = a->ldc; k.ngroups = a->ngroups; k.flags = a->flags;
    k.bsa = a->bsa; k.bsb = a->bsb; k.bsc = a->bsc; k.alpha = a->alpha; k.bias = a->bias; k.skip_if = a->skip_if;
    for (int i = 0; i < a->ngroups; ++i) k.g[i] = a->g[i];
    const int tiles = ((a->M + BM - 1) / BM) * ((a->N + BN - 1) / BN);
    dim3 grid(tiles, a->ngroups * a->nbatch, 1), block(NT, 1, 1);
    switch (a->mode) {
        case 0:
            if constexpr (BM == 128) hipLaunchKernelGGL((gemm_kernel_w3<BM, BN, false, false, BK>), grid, block, 0, st, k);
            else hipLaunchKernelGGL((gemm_kernel<BM, BN, false, false, BK>), grid, block, 0, st, k);
            break;
        case 1:
            if constexpr (BM == 128) hipLaunchKernelGGL((gemm_kernel_w3<BM, BN, false, true, BK>), grid, block, 0, st, k);
            else hipLaunchKernelGGL((gemm_kernel<BM, BN, false, true, BK>), grid, block, 0, st, k);
            break;
        case 2:
            if constexpr (BM == 128) hipLaunchKernelGGL((gemm_kernel_w3<BM, BN, true, true, BK>), grid, block, 0, st, k);
            else hipLaunchKernelGGL((gemm_kernel<BM, BN, true, true, BK>), grid, block, 0, st, k);
            break;
        default: return DLSG_EINVAL;
    }
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}

// out = sum of slabs (+bias) (tanh)
__global__ void slab_reduce_kernel(const float* __restrict__ slabs, int nslab, int64_t stride,
                                   const float* __restrict__ bias, float* __restrict__ out, int64_t rows, int n,
                                   int ldo, int flags) {
    const int64_t total = rows * n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t rrow = i / n;
        const int col = (int)(i - rrow * n);
        float s = 0.f;
        int k = 0;
        for (; k + 4 <= nslab; k += 4) {               // four slabs per round trip; same order of additions
            const float t0 = slabs[k * stride + i], t1 = slabs[(k + 1) * stride + i], t2 = slabs[(k + 2) * stride + i],
                        t3 = slabs[(k + 3) * stride + i];
            s += t0; s += t1; s += t2; s += t3;
        }
        for (; k < nslab; ++k) s += slabs[k * stride + i];
        if ((flags & DLSG_GEMM_BIAS) && bias) s += bias[col];
        float* op = out + rrow * ldo + col;
        if (flags & DLSG_GEMM_ACCUM) s += *op;
        if (flags & DLSG_GEMM_TANH) s = tanhf(s);
        *op = s;
    }
}

}  // namespace

int dlsg_gemm_bf16x3_dispatch(const dlsg_gemm_args* a, hipStream_t st);   // gemm_bf16x3.hip
int dlsg_gemm_big_ok(const dlsg_gemm_args* a);                             // gemm_big.hip
int dlsg_gemm_big_dispatch(const dlsg_gemm_args* a, hipStream_t st, int bn);
int dlsg_gemm_sk_wanted(const dlsg_gemm_args* a);                          // gemm_sk.hip
int dlsg_gemm_sk_dispatch(const dlsg_gemm_args* a, hipStream_t st);

extern "C" int dlsg_abi_version(void) { return DLSG_ABI_VERSION; }

// Which kernel family a call runs on (fp32 path): the ONE place that decides, shared by dlsg_gemm and dlsg_gemm_variant.
// *m1: rows of the head that DLSG_GEMM_V_256_HEAD multiplies on 256 x 256 tiles (the rest goes through the choice again).
static int gemm_plan(const dlsg_gemm_args* a, int64_t* m1_out) {
    const int64_t z = (int64_t)a->ngroups * a->nbatch;
    const int64_t tilesL = (int64_t)((a->M + 127) / 128) * ((a->N + 127) / 128) * z;
    if (a->flags & DLSG_GEMM_SK) return DLSG_GEMM_V_SK;
    if (a->flags & DLSG_GEMM_TILE256) return (a->flags & DLSG_GEMM_FORCE128) ? DLSG_GEMM_V_256x128 : DLSG_GEMM_V_256;
    if ((a->flags & (DLSG_GEMM_FORCE64 | DLSG_GEMM_FORCE128)) == (DLSG_GEMM_FORCE64 | DLSG_GEMM_FORCE128))
        return DLSG_GEMM_V_128x64;                 // both bits: the 128 x 64 tile
    if (a->flags & DLSG_GEMM_FORCE64) return DLSG_GEMM_V_64;
    if (a->flags & DLSG_GEMM_FORCE128) return DLSG_GEMM_V_128;
    // M <= 128, row-major A (NT / NN): weight-streaming recurrent products -> skinny kernels (64- or 128-row tiles)
    if (a->M <= 128 && a->mode != 2 && a->N >= 64) return DLSG_GEMM_V_SKINNY;
    // The caller brought a workspace: the products that keep every CU busy long enough run as ONE persistent stream-K launch
    // (gemm_sk.hip: no partly filled last round, no second launch for remaining rows, no slab fold for deep contractions; the rule
    // and the measurements behind it are in dlsg_gemm_sk_wanted)
    if (dlsg_gemm_sk_wanted(a)) return DLSG_GEMM_V_SK;
    // measured on MI355X (tools/archive/gemm_bench.py): the 128x128 tile only wins once it fills the chip several times over
    // (Wave quantisation is not what the 128-tile launches lose: giving that kernel whole 768-slot rounds only and the remaining
    // row panels to the 64-tile kernel was measured 1-3 % SLOWER on the region projection (4.33 rounds) and on the deep weight
    // gradient (2.67 rounds), tools/archive/gemm_split_probe.py -- workgroups of a partly filled last round simply run faster.)
    if (tilesL >= 1000) {
        // 256 x 256 tiles (gemm_big.hip: one workgroup per CU, 64 flop per staged byte) for as many row panels as come in whole
        // rounds of the 256 CUs -- a launch that leaves its last round mostly empty loses more than the tile gains (region
        // projection as one launch of 832 tiles = 3.25 rounds: 108 TFLOP/s against 122 on the 128 x 128 tile, 135-142 per full
        // round) -- and the remaining rows through this choice again.  tools/archive/gemm_vs_rocblas.py, tools/gemm_census.py.
        // (groups of different widths -- the decoder's weight-gradient blocks 4096 x {1024 x 10, 300} -- stay on the 128 tile: eight
        // of the full-width groups as two rounds of 256 tiles and the rest behind them measured 1 313 us against 1 305)
        bool plain = dlsg_gemm_big_ok(a) != 0;
        for (int i = 0; i < a->ngroups && plain; ++i) plain = a->g[i].N == 0 || a->g[i].N == a->N;
        const int64_t pad_n = (a->N + 255) / 256 * 256;
        if (plain && pad_n * 10 <= (int64_t)a->N * 11) {
            const int64_t panel = (pad_n / 256) * z;                   // tiles per 256-row panel
            int64_t g = panel, b = 256;
            while (b) { const int64_t t = g % b; g = b; b = t; }       // gcd(panel, 256)
            const int64_t need = 256 / g;                              // row panels per whole number of rounds
            const int64_t m1 = (a->M / 256) / need * need * 256;
            if (m1 == a->M) return DLSG_GEMM_V_256;
            if (m1 > 0 && (a->M - m1) * 4 <= a->M) {                   // whole rounds cover nearly everything
                if (m1_out) *m1_out = m1;
                return DLSG_GEMM_V_256_HEAD;
            }
        }
        return DLSG_GEMM_V_128;
    }
    // Mid-size launches (tools/archive/gemm_tile_probe.py; M = 1664 = 26 frames x 64 clips and the weight gradients over them), when
    // the 128-row panels waste < 10 % of their rows: the TN form takes the 128 x 128 tile from 500 tiles (2048 x 2048 x 1664 x 3
    // TN: 402 us against 445; NT / NN only from 1000), between 200 tiles and that a 128 x 64 tile (21 flop per staged byte instead of 16, twice the
    // workgroups of the square tile) wins 3-8 % over the 64 x 64 one; below that only the small tile fills the chip.
    const int padM = (a->M + 127) / 128 * 128;
    if ((padM - a->M) * 10 <= a->M) {
        // (500-999 tiles: only the TN form gains from the square tile -- 2048 x 2048 x 1664 x 3: 343 us against 354; the BiLSTM's
        // input projection NT 1664 x 4096 x 1024 x 2 runs 264 us on 128 x 64 against 284, tools/archive/gemm_mid_probe.py)
        if (tilesL >= 500 && a->mode == 2) return DLSG_GEMM_V_128;       // (1024 x 1024 x 512 x 8 groups: 82 us against 100)
        if (tilesL >= 200) return DLSG_GEMM_V_128x64;
    }
    return DLSG_GEMM_V_64;
}

static bool gemm_args_ok(const dlsg_gemm_args* a) {
    return a && a->ngroups >= 1 && a->ngroups <= DLSG_GEMM_MAXG && a->nbatch >= 1 && a->M >= 0 && a->N >= 0 &&
           (int64_t)a->ngroups * a->nbatch <= 65535;
}

extern "C" int dlsg_gemm_variant(const dlsg_gemm_args* a) {
    if (!gemm_args_ok(a)) return DLSG_EINVAL;
    if (a->flags & DLSG_GEMM_BF16X3) return DLSG_EINVAL;       // (the split-bf16 path has its own tile rule, gemm_bf16x3.hip)
    return gemm_plan(a, nullptr);
}

extern "C" int dlsg_gemm(const dlsg_gemm_args* a, void* stream) {
    if (!gemm_args_ok(a)) return DLSG_EINVAL;
    if (a->M == 0 || a->N == 0) return DLSG_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (a->flags & DLSG_GEMM_BF16X3) return dlsg_gemm_bf16x3_dispatch(a, st);
    int64_t m1 = 0;
    switch (gemm_plan(a, &m1)) {
        case DLSG_GEMM_V_SK: return dlsg_gemm_sk_dispatch(a, st);
        case DLSG_GEMM_V_256: return dlsg_gemm_big_dispatch(a, st, 256);
        case DLSG_GEMM_V_256x128: return dlsg_gemm_big_dispatch(a, st, 128);
        case DLSG_GEMM_V_256_HEAD: {
            dlsg_gemm_args head = *a, tail = *a;
            head.M = (int)m1;
            tail.M = a->M - (int)m1;
            for (int i = 0; i < a->ngroups; ++i) {
                const int64_t ldc = a->g[i].ldc ? a->g[i].ldc : (int64_t)a->ldc;
                tail.g[i].A = a->g[i].A + (a->mode == 2 ? m1 : m1 * a->g[i].lda);
                tail.g[i].C = a->g[i].C + m1 * ldc;
            }
            const int rc = dlsg_gemm_big_dispatch(&head, st, 256);
            return rc != DLSG_OK ? rc : dlsg_gemm(&tail, stream);
        }
        case DLSG_GEMM_V_SKINNY: return launch_skinny(a, st);
        case DLSG_GEMM_V_128: return launch<128, 128, 32>(a, st);
        case DLSG_GEMM_V_128x64: return launch<128, 64, 64>(a, st);
        default: return launch<64, 64, 64>(a, st);
    }
}

extern "C" int dlsg_slab_reduce(const float* slabs, int nslab, int64_t slab_stride, const float* bias, float* out,
                                int64_t rows, int n, int ldo, int flags, void* stream) {
    if (!slabs || !out || nslab < 1) return DLSG_EINVAL;
    const int64_t total = rows * n;
    if (total == 0) return DLSG_OK;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), slabs,
                       nslab, slab_stride, bias, out, rows, n, ldo, flags);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
