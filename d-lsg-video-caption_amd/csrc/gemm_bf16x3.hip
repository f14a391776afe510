// Split-bf16 ("bf16x3") variant of the grouped GEMM: fp32 operands in HBM, bf16 matrix cores inside.
//
// Each fp32 value is split while it is staged through registers: x = hi + lo with hi = bf16(x), lo = bf16(x - hi)
// (x - hi is exact in fp32).  The product keeps the three leading terms, accumulated in fp32 on
// v_mfma_f32_32x32x16_bf16:      a*b ~= a_lo*b_hi + a_hi*b_lo + a_hi*b_hi          (dropped: a_lo*b_lo <= 2^-18 |ab|)
// Relative error per product ~1e-5 (fp32: 6e-8), against 16x the fp32 matrix rate per instruction => 3 bf16 MFMAs
// (96 cycles) replace 8 fp32 MFMAs (512 cycles) per 32x32x16 block.  Selected per call with DLSG_GEMM_BF16X3; parity of
// the whole model with it enabled is asserted in tests/test_gpu_parity.py (logits <= 1e-3, token ids bit-exact).
//
// Structure = gemm.hip: 256 threads as 2x2 waves, BMxBN block tile (64x64 with BK = 64, 128x128 with BK = 32: the 64-deep
// K tile halves the barriers per MFMA and is worth +25-35 % on the model's shapes), global -> registers -> LDS with the next
// K tile in flight.  LDS holds four bf16 planes (A_hi, A_lo, B_hi, B_lo):
//   * k-contiguous operands : [row][32 + 8] (80-byte rows: 16-B slot stride 5 mod 16, conflict-free ds_read_b128)
//   * m/n-contiguous operands (NN / TN): kept k-major, [k][rows + 32], written with 8-byte stores of 4 consecutive
//     rows and read back through the hardware transpose read ds_read_b64_tr_b16 (4 k x 16 rows per 16-lane group);
//     the row stride (rows+32 bf16 = 16 mod 64 dwords) keeps the four k rows of a read on disjoint banks.
// The split uses v_cvt_pk_bf16_f32 (round-to-nearest-even): 6 VALU instructions per two elements.
#include "common.hpp"
#include "dlsg.h"

namespace {

constexpr int NT = 256;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short short8 __attribute__((ext_vector_type(8)));

struct KArgs {
    int M, N, ldc, ngroups, flags;
    int64_t bsa, bsb, bsc;
    float alpha;
    const float* bias;
    const int32_t* skip_if;
    dlsg_gemm_group g[DLSG_GEMM_MAXG];
};

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short short4v __attribute__((ext_vector_type(4)));

// split two floats into packed hi / lo bf16 pairs (element 0 in the low half); v_cvt_pk_bf16_f32 rounds to nearest even
__device__ __forceinline__ void split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    const f32x2 v = {x0, x1};
    const uint32_t h = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
    const f32x2 rres = {x0 - __uint_as_float(h << 16), x1 - __uint_as_float(h & 0xFFFF0000u)};
    hi = h;
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(rres, bf16x2));
}

template <int ROWS, bool T, int BK>
struct Geom {
    static constexpr int NV = ROWS * BK / 4 / NT;                     // float4 per thread per tile
    static constexpr int LDP = BK + 8;                                // k-contiguous row stride (bf16): 16-B slot stride odd
    static constexpr int LDT = ROWS + 32;                             // k-major row stride (bf16)
    static constexpr int PLANE = T ? BK * LDT : ROWS * LDP;           // bf16 elements per plane
    static constexpr int KQ = BK / 4;                                 // float4 per row of a k-contiguous tile
};

// T=false: (row,k) at base[row*ld + k], one float4 = 4 consecutive k of a row.
// T=true : (row,k) at base[k*ld + row], one float4 = 4 consecutive rows at one k.
// Interior tiles take the branch-free path (see gemm.hip load_tile_fast); edge tiles / unaligned operands the guarded one.
template <int ROWS, bool T, int BK>
__device__ __forceinline__ void load_tile(const float* __restrict__ base, int64_t ld, int row0, int k0, int rmax, int K,
                                          bool vec_ok, f32x4 (&regs)[Geom<ROWS, T, BK>::NV]) {
    constexpr int NV = Geom<ROWS, T, BK>::NV;
    constexpr int KQ = Geom<ROWS, T, BK>::KQ;
    const bool fast = vec_ok && (k0 + BK <= K) && (T ? (row0 + ROWS <= rmax) : (rmax > 0));
    if (fast) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int f = threadIdx.x + NT * j;
            const float* ptr;
            if (!T) ptr = base + (int64_t)min(row0 + f / KQ, rmax - 1) * ld + k0 + 4 * (f % KQ);
            else ptr = base + (int64_t)(k0 + f / (ROWS / 4)) * ld + row0 + 4 * (f % (ROWS / 4));
            regs[j] = *reinterpret_cast<const f32x4*>(ptr);
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int f = threadIdx.x + NT * j;
        int gr, gk, nvalid;
        const float* ptr;
        if (!T) {
            const int row = f / KQ, kq = f % KQ;
            gr = row0 + row; gk = k0 + 4 * kq;
            ptr = base + (int64_t)gr * ld + gk;
            nvalid = (gr < rmax) ? min(max(K - gk, 0), 4) : 0;
        } else {
            const int k = f / (ROWS / 4), mq = f % (ROWS / 4);
            gk = k0 + k; gr = row0 + 4 * mq;
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
__device__ __forceinline__ void store_tile(unsigned short* __restrict__ hi_p, unsigned short* __restrict__ lo_p,
                                           const f32x4 (&regs)[Geom<ROWS, T, BK>::NV]) {
    constexpr int NV = Geom<ROWS, T, BK>::NV;
    constexpr int KQ = Geom<ROWS, T, BK>::KQ;
    constexpr int LDP = Geom<ROWS, T, BK>::LDP;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int f = threadIdx.x + NT * j;
        const int off = T ? (f / (ROWS / 4)) * Geom<ROWS, T, BK>::LDT + 4 * (f % (ROWS / 4)) : (f / KQ) * LDP + 4 * (f % KQ);
        uint32_t h0, l0, h1, l1;
        split2(regs[j][0], regs[j][1], h0, l0);
        split2(regs[j][2], regs[j][3], h1, l1);
        *reinterpret_cast<uint2*>(hi_p + off) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(lo_p + off) = make_uint2(l0, l1);
    }
}

// MFMA 32x32x16 operand fragment of the 32-row subtile starting at `rowbase`, chunk c (k = 16c + 8h + j, j < 8)
template <int ROWS, bool T, int BK>
__device__ __forceinline__ bf16x8 ld_frag(const unsigned short* __restrict__ plane, int rowbase, int c, int lane) {
    constexpr int LDP = Geom<ROWS, T, BK>::LDP;
    if (!T) {
        const int r = lane & 31, h = lane >> 5;
        const short8 v = *reinterpret_cast<const short8*>(plane + (rowbase + r) * LDP + 16 * c + 8 * h);
        return __builtin_bit_cast(bf16x8, v);
    } else {
        // 16-lane group g: rows rowbase + 16*(g&1) + i, k half h = g>>1.  Lane 4q+p of a group addresses k row q,
        // rows 4p..4p+3 of the 4 x 16 block; it receives the 4 k values of its own row (i = lane & 15).
        constexpr int LDT = Geom<ROWS, T, BK>::LDT;
        const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
        const int kb = 16 * c + 8 * (g >> 1);
        const unsigned short* a0 = plane + (kb + q) * LDT + rowbase + 16 * (g & 1) + 4 * pp;
        typedef short4v __attribute__((address_space(3))) * lds_s4;
        const short4v lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(a0));
        const short4v hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(a0 + 4 * LDT));
        const short8 v = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
        return __builtin_bit_cast(bf16x8, v);
    }
}

template <int BM, int BN, bool AT, bool BT, int BK>
__device__ __forceinline__ void gemm_x3_body(const KArgs& p) {
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int TM = WM / 32, TN = WN / 32;
    using GA = Geom<BM, AT, BK>;
    using GB = Geom<BN, BT, BK>;
    __shared__ __attribute__((aligned(16))) unsigned short lds[2 * GA::PLANE + 2 * GB::PLANE];
    unsigned short* a_hi = lds;
    unsigned short* a_lo = lds + GA::PLANE;
    unsigned short* b_hi = lds + 2 * GA::PLANE;
    unsigned short* b_lo = lds + 2 * GA::PLANE + GB::PLANE;

    const int tiles_n = (p.N + BN - 1) / BN;
    int tm, z, tn;
    dlsg::gemm_tile_map(tiles_n, tm, z, tn);      // XCD-aware tile order (common.hpp)
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
        store_tile<BM, AT, BK>(a_hi, a_lo, ra);
        store_tile<BN, BT, BK>(b_hi, b_lo, rb);
        __syncthreads();
        if (kt + 1 < nk) {
            load_tile<BM, AT, BK>(A, grp.lda, m0, (kt + 1) * BK, p.M, K, vecA, ra);
            load_tile<BN, BT, BK>(B, grp.ldb, n0, (kt + 1) * BK, Ng, K, vecB, rb);
        }
#pragma unroll
        for (int c = 0; c < BK / 16; ++c) {    // 16-deep MFMA chunks of the K tile; lane half h owns k = 16c + 8h + j
            bf16x8 fah[TM], fal[TM], fbh[TN], fbl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                fah[i] = ld_frag<BM, AT, BK>(a_hi, wm * WM + i * 32, c, lane);
                fal[i] = ld_frag<BM, AT, BK>(a_lo, wm * WM + i * 32, c, lane);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                fbh[j] = ld_frag<BN, BT, BK>(b_hi, wn * WN + j * 32, c, lane);
                fbl[j] = ld_frag<BN, BT, BK>(b_lo, wn * WN + j * 32, c, lane);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[i], fbh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fbl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fbh[j], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }

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
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row >= p.M) continue;
                float* cp = C + (int64_t)row * (grp.ldc ? grp.ldc : (int64_t)p.ldc) + col;
                float v = p.alpha * acc[i][j][e] + bv;
                if (accum) v += *cp;
                if (do_tanh) v = tanhf(v);
                *cp = v;
            }
        }
}

// Two entry points over one body: the 128 x 128 tile is compiled for 3 waves per SIMD (<= 170 registers; the compiler's
// own choice is 182-196, i.e. two workgroups per CU): measured +7 % on the region projection and +11 % on deep TN
// products (tools/archive/gemm_fp32_probe.py); the 64 x 64 tile already fits 3-4 waves and keeps the compiler's allocation.
template <int BM, int BN, bool AT, bool BT, int BK>
__global__ __launch_bounds__(NT) void gemm_x3_kernel(const KArgs p) { gemm_x3_body<BM, BN, AT, BT, BK>(p); }
template <int BM, int BN, bool AT, bool BT, int BK>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(3, 3))) void gemm_x3_kernel_w3(const KArgs p) { gemm_x3_body<BM, BN, AT, BT, BK>(p); }

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
            if constexpr (BM == 128) hipLaunchKernelGGL((gemm_x3_kernel_w3<BM, BN, false, false, BK>), grid, block, 0, st, k);
            else hipLaunchKernelGGL((gemm_x3_kernel<BM, BN, false, false, BK>), grid, block, 0, st, k);
            break;
        case 1:
            if constexpr (BM == 128) hipLaunchKernelGGL((gemm_x3_kernel_w3<BM, BN, false, true, BK>), grid, block, 0, st, k);
            else hipLaunchKernelGGL((gemm_x3_kernel<BM, BN, false, true, BK>), grid, block, 0, st, k);
            break;
        case 2:
            if constexpr (BM == 128) hipLaunchKernelGGL((gemm_x3_kernel_w3<BM, BN, true, true, BK>), grid, block, 0, st, k);
            else hipLaunchKernelGGL((gemm_x3_kernel<BM, BN, true, true, BK>), grid, block, 0, st, k);
            break;
        default: return DLSG_EINVAL;
    }
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}

// ------------------------------------------------------------------------------------------------ skinny (M <= 64) x3
// Same structure as skinny_kernel in gemm.hip (4 waves split K in 32-wide chunks on one 64 x 32 tile, operands straight
// to registers, partial tiles summed through LDS); the 16 fp32 values a lane holds per operand row are split in
// registers into two 8-element hi/lo fragments (k = k0 + 16h + 8c + j) and fed to 12 bf16 MFMAs per chunk.
__device__ __forceinline__ void split8(const float* x, bf16x8& hi, bf16x8& lo) {
    uint32_t h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split2(x[2 * i], x[2 * i + 1], h[i], l[i]);
    const uint4 hv = make_uint4(h[0], h[1], h[2], h[3]), lv = make_uint4(l[0], l[1], l[2], l[3]);
    hi = __builtin_bit_cast(bf16x8, hv);
    lo = __builtin_bit_cast(bf16x8, lv);
}

// LDS-staged 64 x 32 skinny kernel: same structure as skinny_kernel in gemm.hip, bf16x3 MFMAs (split in registers)
constexpr int SK = 128;            // super-chunk depth
constexpr int SLD = SK + 4;        // LDS row stride (floats): 16-B slot stride 33 = 1 mod 16

template <bool BT>
__global__ __launch_bounds__(NT) void skinny_x3_kernel(const KArgs p) {
    __shared__ __attribute__((aligned(16))) float lds[(64 + 32) * SLD];     // 50,688 B; reused for the final reduction
    float* ldsA = lds;
    float* ldsB = lds + 64 * SLD;
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

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    f32x4 ra[8], rb[4];          // staged operands of the next super-chunk
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
            for (int j = 0; j < 8; ++j) {
                const int f = threadIdx.x + NT * j;
                const int row = min(f >> 5, M - 1);       // rows >= M feed output rows that are never stored
                ra[j] = *reinterpret_cast<const f32x4*>(A + (int64_t)row * grp.lda + k0 + 4 * (f & 31));
            }
        } else if (vecA && (K & 3) == 0 && K >= 4) {       // aligned tail: whole float4s, clamped address + select
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int f = threadIdx.x + NT * j;
                const int row = min(f >> 5, M - 1), k = k0 + 4 * (f & 31);
                const f32x4 v = *reinterpret_cast<const f32x4*>(A + (int64_t)row * grp.lda + min(k, K - 4));
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                ra[j] = (k < K) ? v : zero;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
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
        for (int j = 0; j < 8; ++j) {
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
        float fa[2][16];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
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
        // lane half h holds k = 32w + 16h + s, s < 16: two 8-deep bf16 MFMA chunks, split in registers
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            bf16x8 bh, bl, ah[2], al[2];
            split8(&bcur[8 * cc], bh, bl);
            split8(&fa[0][8 * cc], ah[0], al[0]);
            split8(&fa[1][8 * cc], ah[1], al[1]);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], bh, acc[mi], 0, 0, 0);
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bl, acc[mi], 0, 0, 0);
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bh, acc[mi], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // ---- sum the 4 waves' partial tiles through LDS (staging buffers are free now); wave w finalises e in [4w, 4w+4)
    float* red = lds;               // [4][2][16][64] floats = 32 KB
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int e = 0; e < 16; ++e) red[((w * 2 + mi) * 16 + e) * 64 + lane] = acc[mi][e];
    __syncthreads();
    const float* biasp = grp.bias ? grp.bias : p.bias;
    const bool accum = p.flags & DLSG_GEMM_ACCUM, use_bias = (p.flags & DLSG_GEMM_BIAS) && biasp != nullptr;
    const bool do_tanh = p.flags & DLSG_GEMM_TANH;
    if (col < N) {
        const float bv = use_bias ? biasp[col] : 0.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ee = 0; ee < 4; ++ee) {
                const int e = 4 * w + ee;
                const int row = 32 * mi + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row >= M) continue;
                float v = 0.f;
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) v += red[((ww * 2 + mi) * 16 + e) * 64 + lane];
                v = p.alpha * v + bv;
                float* cp = C + (int64_t)row * (grp.ldc ? grp.ldc : (int64_t)p.ldc) + col;
                if (accum) v += *cp;
                if (do_tanh) v = tanhf(v);
                *cp = v;
            }
    }
}

int launch_skinny_x3(const dlsg_gemm_args* a, hipStream_t st) {
    KArgs k;
    k.M = a->M; k.N = a->N; k.ldc = a->ldc; k.ngroups = a->ngroups; k.flags = a->flags;
    k.bsa = a->bsa; k.bsb = a->bsb; k.bsc = a->bsc; k.alpha = a->alpha; k.bias = a->bias; k.skip_if = a->skip_if;
    for (int i = 0; i < a->ngroups; ++i) k.g[i] = a->g[i];
    dim3 grid((a->N + 31) / 32, a->ngroups * a->nbatch, 1), block(NT, 1, 1);
    if (a->mode == 0) hipLaunchKernelGGL((skinny_x3_kernel<false>), grid, block, 0, st, k);
    else hipLaunchKernelGGL((skinny_x3_kernel<true>), grid, block, 0, st, k);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}

}  // namespace

// called from dlsg_gemm (gemm.hip) when DLSG_GEMM_BF16X3 is set
int dlsg_gemm_bf16x3_dispatch(const dlsg_gemm_args* a, hipStream_t st) {
    const int64_t z = (int64_t)a->ngroups * a->nbatch;
    const int64_t tilesL = (int64_t)((a->M + 127) / 128) * ((a->N + 127) / 128) * z;
    if (a->flags & DLSG_GEMM_FORCE64) return launch<64, 64, 64>(a, st);
    if (a->flags & DLSG_GEMM_FORCE128) return launch<128, 128, 32>(a, st);
    if (a->M <= 64 && a->mode != 2 && a->N >= 64) return launch_skinny_x3(a, st);
    // measured (tools/archive/gemm_bench.py): the 128x128 tile wins only once it fills the chip several times over
    if (tilesL >= 1000) return launch<128, 128, 32>(a, st);
    return launch<64, 64, 64>(a, st);
}
