// Blocks of the DiscV2 critic's schedule (dlsg_amd/critic.py; models/model.py:110-168, layer.py:661-715,
// run_gun.py:339-398).  A WGAN-GP critic update differentiates the critic twice; every block here exists at the three levels
// include/dlsg.h describes: fwd, bwd (vector-Jacobian product) and bwd2 = the derivative of (fwd, bwd) along a tangent of the
// block's input.  The bwd2 kernels are the fwd and bwd lines differentiated by hand, line by line ("xd" = derivative of "x").
//
// Sizes: a critic update at batch 64 scores 192 captions of 26 words, 512 channels -- every block is far below a millisecond of
// arithmetic, so the kernels are organised for FEW LAUNCHES and plain coalesced access, not for the matrix pipe: one workgroup of
// 256 threads per caption (thread t owns channels 2t, 2t+1: 8-byte accesses, a 512-wide row per workgroup access), the per-caption
// L x L / L x T matrices live in LDS, contractions over the 512 channels are per-thread partial sums folded by DPP wave
// reductions, contractions over the words run per channel out of registers.  The dense products between the blocks are dlsg_gemm.
#include "common.hpp"
#include "dlsg.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int C = DLSG_CRIT_C, LMAX = DLSG_CRIT_LMAX, TMAX = DLSG_CRIT_TMAX, NT = 256;
inline hipStream_t ST(void* s) { return reinterpret_cast<hipStream_t>(s); }

__device__ __forceinline__ f32x2 ld2(const float* p) { return *reinterpret_cast<const f32x2*>(p); }
__device__ __forceinline__ void st2(float* p, f32x2 v) { *reinterpret_cast<f32x2*>(p) = v; }
__device__ __forceinline__ float dot2(f32x2 a, f32x2 b) { return a.x * b.x + a.y * b.y; }

// ---------------------------------------------------------------------------------------------- block-wide helpers (256 threads)
// Per-caption matrices live in LDS, zero-filled once (zero_lds), with row strides that are multiples of 4 floats: the helpers read
// whole padded rows / columns without bounds tests (a test per element puts a branch between the LDS reads and every read then
// waits out its own latency: 30+ us per product instead of ~3).
constexpr int LDW = LMAX + 4;        // L x L matrices
constexpr int LDT = TMAX + 4;        // L x T matrices
constexpr int RED = 4 * 16 * 64;     // floats of LDS a pairdot needs to fold its four partial tiles

__device__ __forceinline__ void zero_lds(float* m, int n) {
    for (int k = threadIdx.x; k < n; k += NT) m[k] = 0.f;
}

// out[i * ldo + j] = alpha * sum_c A[i][c] B[j][c], i < RA <= 32, j < RB <= 32: contraction over the 512 channels as ONE 32 x 32 tile
// of v_mfma_f32_32x32x2_f32 (exact fp32).  Wave w contracts channels [128 w, 128 w + 128): lane (r = lane & 31, h = lane >> 5) holds
// A[r][128 w + 64 h + s] and B[r][...] for s < 64 (16 x 16-byte loads each, all in flight together); MFMA s contracts the channel
// pair (128 w + s, 128 w + 64 + s) -- the same pair for both operands.  The four partial tiles meet in LDS (`red`, RED floats).
// A, B global rows (strides lda, ldb, 16-byte aligned).  Ends with a barrier: `out` (LDS) is ready for every thread.
__device__ __forceinline__ void pairdot(const float* __restrict__ A, int64_t lda, int RA, const float* __restrict__ B, int64_t ldb, int RB,
                                        float alpha, float* out, int ldo, float* red) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
    f32x4 a[16], b[16];
    const float* ap = A + (int64_t)r * lda + 128 * w + 64 * h;
    const float* bp = B + (int64_t)r * ldb + 128 * w + 64 * h;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        a[q] = r < RA ? *reinterpret_cast<const f32x4*>(ap + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
        b[q] = r < RB ? *reinterpret_cast<const f32x4*>(bp + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][e], b[q][e], acc, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 16; ++e) red[(w * 16 + e) * 64 + lane] = acc[e];
    __syncthreads();
    // C / D map of the 32 x 32 tile: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
    for (int k = tid; k < 16 * 64; k += NT) {
        const int e = k >> 6, ln = k & 63, i = (e & 3) + 8 * (e >> 2) + 4 * (ln >> 5), jj = ln & 31;
        if (i < RA && jj < RB) out[i * ldo + jj] = alpha * ((red[k] + red[1024 + k]) + (red[2048 + k] + red[3072 + k]));
    }
    __syncthreads();
}

// dots[i] = sum_c X[i][c] v[c] for i < R <= LMAX (X global rows, v this thread's channel pair); all rows are requested before the
// first reduction.  Result in LDS `out`; scratch: 4 * LMAX floats.
__device__ __forceinline__ void rowdots(const float* __restrict__ X, int64_t ldx, int R, f32x2 v, float* out, float* scratch) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    f32x2 x[LMAX];
#pragma unroll
    for (int i = 0; i < LMAX; ++i) x[i] = i < R ? ld2(X + (int64_t)i * ldx + 2 * tid) : f32x2{0.f, 0.f};
#pragma unroll
    for (int i = 0; i < LMAX; ++i) {
        const float s = dlsg::wave_sum(dot2(x[i], v));
        if (lane == 0) scratch[w * LMAX + i] = s;
    }
    __syncthreads();
    if (tid < R) out[tid] = (scratch[tid] + scratch[LMAX + tid]) + (scratch[2 * LMAX + tid] + scratch[3 * LMAX + tid]);
    __syncthreads();
}

// block-wide sum of one value per thread; every thread gets it.  red: 4 floats of LDS, reusable after the call.
__device__ __forceinline__ float bsum(float v, float* red) {
    v = dlsg::wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// Y[i][c] (=|+=) alpha * (sum_j M(i, j) X[j][c] + sum_j M2(i, j) X2[j][c]), i < RI, j < RJ <= RJM: contraction over words / proposals
// per channel pair out of registers.  M, M2 zero-padded LDS matrices: M(i, j) = M[i * ldm + j] (ldm % 4 == 0, whole padded rows read
// as 16-byte vectors), or M[j * ldm + i] when TR.  X rows global; DUAL: the second product is present.
template <int RJM, bool TR, bool DUAL>
__device__ __forceinline__ void chanprod(const float* M, int ldm, const float* __restrict__ X, int64_t ldx, const float* M2,
                                         const float* __restrict__ X2, int64_t ldx2, int RI, int RJ, float alpha, float* __restrict__ Y,
                                         int64_t ldy, bool accum) {
    const int tid = threadIdx.x;
    f32x2 x[RJM], x2[DUAL ? RJM : 1];
#pragma unroll
    for (int j = 0; j < RJM; ++j) {
        x[j] = j < RJ ? ld2(X + (int64_t)j * ldx + 2 * tid) : f32x2{0.f, 0.f};
        if (DUAL) x2[j] = j < RJ ? ld2(X2 + (int64_t)j * ldx2 + 2 * tid) : f32x2{0.f, 0.f};
    }
    constexpr int IC = 8;               // rows of Y whose old values (accum) are fetched together, ahead of their use
    f32x2 yo[IC];
    for (int i = 0; i < RI; ++i) {
        if (accum && (i % IC) == 0) {
            // Y += : a load -> add -> store per row is a memory round trip per row (26 of them in a row of words); the old values
            // of the next IC rows are requested at once
#pragma unroll
            for (int u = 0; u < IC; ++u) yo[u] = (i + u < RI) ? ld2(Y + (int64_t)(i + u) * ldy + 2 * tid) : f32x2{0.f, 0.f};
        }
        float m[RJM], m2[DUAL ? RJM : 1];
        if (TR) {
#pragma unroll
            for (int j = 0; j < RJM; ++j) {
                m[j] = M[j * ldm + i];
                if (DUAL) m2[j] = M2[j * ldm + i];
            }
        } else {
#pragma unroll
            for (int j = 0; j < RJM; j += 4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(M + i * ldm + j);
                m[j] = v[0]; m[j + 1] = v[1]; m[j + 2] = v[2]; m[j + 3] = v[3];
                if (DUAL) {
                    const f32x4 v2 = *reinterpret_cast<const f32x4*>(M2 + i * ldm + j);
                    m2[j] = v2[0]; m2[j + 1] = v2[1]; m2[j + 2] = v2[2]; m2[j + 3] = v2[3];
                }
            }
        }
        f32x2 acc = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < RJM; ++j) {
            acc.x += m[j] * x[j].x; acc.y += m[j] * x[j].y;
            if (DUAL) { acc.x += m2[j] * x2[j].x; acc.y += m2[j] * x2[j].y; }
        }
        float* y = Y + (int64_t)i * ldy + 2 * tid;
        f32x2 r = {alpha * acc.x, alpha * acc.y};
        if (accum) {
            f32x2 o = yo[0];
#pragma unroll
            for (int u = 1; u < IC; ++u) o = (i % IC) == u ? yo[u] : o;
            r.x += o.x; r.y += o.y;
        }
        st2(y, r);
    }
}

__device__ __forceinline__ uint64_t seed_of(uint64_t seed, const uint64_t* seed_ptr) { return seed + (seed_ptr ? *seed_ptr : 0ull); }

// ================================================================================================ vocabulary projection glue
__global__ __launch_bounds__(128) void embed_mix_kernel(const float* __restrict__ proj_tm, const int64_t* __restrict__ ids,
                                                        const float* __restrict__ W, const float* __restrict__ bias,
                                                        const float* __restrict__ eps, float* __restrict__ h, int ng, int B, int L, int V) {
    const int r = blockIdx.x, b = r / L, l = r - b * L, c = 4 * threadIdx.x;
    const f32x4 bi = *reinterpret_cast<const f32x4*>(bias + c);
    const f32x4 pf = *reinterpret_cast<const f32x4*>(proj_tm + ((int64_t)l * B + b) * C + c);
    const f32x4 hf = pf + bi;
    const int64_t R = (int64_t)B * L;
    if (ng == 1) {
        *reinterpret_cast<f32x4*>(h + (int64_t)r * C + c) = hf;
        return;
    }
    const int64_t id = ids[r];
    f32x4 hr;
#pragma unroll
    for (int k = 0; k < 4; ++k) hr[k] = W[(int64_t)(c + k) * V + id] + bi[k];
    const float e = eps[b];
    *reinterpret_cast<f32x4*>(h + (int64_t)r * C + c) = hr;
    *reinterpret_cast<f32x4*>(h + (R + r) * C + c) = hf;
    *reinterpret_cast<f32x4*>(h + (2 * R + r) * C + c) = e * hr + (1.f - e) * hf;
}

__global__ __launch_bounds__(128) void embed_mix_bwd_kernel(const float* __restrict__ ch, const float* __restrict__ eps,
                                                            float* __restrict__ dhr, float* __restrict__ dhf_tm, int ng, int B, int L) {
    const int r = blockIdx.x, b = r / L, l = r - b * L, c = 4 * threadIdx.x;
    const int64_t R = (int64_t)B * L;
    float* df = dhf_tm + ((int64_t)l * B + b) * C + c;
    if (ng == 1) {
        *reinterpret_cast<f32x4*>(df) = *reinterpret_cast<const f32x4*>(ch + (int64_t)r * C + c);
        return;
    }
    const f32x4 c0 = *reinterpret_cast<const f32x4*>(ch + (int64_t)r * C + c);
    const f32x4 c1 = *reinterpret_cast<const f32x4*>(ch + (R + r) * C + c);
    const f32x4 c2 = *reinterpret_cast<const f32x4*>(ch + (2 * R + r) * C + c);
    const float e = eps[b];
    *reinterpret_cast<f32x4*>(dhr + (int64_t)r * C + c) = c0 + e * c2;
    *reinterpret_cast<f32x4*>(df) = c1 + (1.f - e) * c2;
}

// dW[c, ids[r]] += dhr[r, c]: the FIRST row that carries an id owns it and adds every row of that id, in row order (no atomics,
// bit-reproducible).  grid (rows, C / 64): a workgroup owns 64 channels; its 4 waves take the matching rows in turn (row lists of
// up to several hundred rows for <pad>), partial sums meet in LDS in a fixed order.
__global__ __launch_bounds__(256) void vocab_scatter_kernel(const float* __restrict__ dhr, const int64_t* __restrict__ ids,
                                                            float* __restrict__ dW, int rows, int V) {
    extern __shared__ int sm_i[];
    int* list = sm_i;                                   // matching rows, in order
    __shared__ int cnt;
    __shared__ int first;
    __shared__ float red[4][64];
    const int r0 = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t id = ids[r0];
    if (tid == 0) { cnt = 0; first = 1; }
    __syncthreads();
    for (int r = tid; r < r0; r += 256)
        if (ids[r] == id) first = 0;
    __syncthreads();
    if (!first) return;
    // ordered list of the rows r >= r0 with this id: chunks of 256 rows, ballot-compacted in order
    for (int base = r0; base < rows; base += 256) {
        const int r = base + tid;
        const bool hit = r < rows && ids[r] == id;
        const uint64_t m = __ballot(hit);
        __shared__ int wcount[4];
        if (lane == 0) wcount[w] = __popcll(m);
        __syncthreads();
        int off = cnt;
        for (int k = 0; k < w; ++k) off += wcount[k];
        if (hit) list[off + __popcll(m & ((1ull << lane) - 1ull))] = r;
        __syncthreads();
        if (tid == 0) cnt += wcount[0] + wcount[1] + wcount[2] + wcount[3];
        __syncthreads();
    }
    const int n = cnt, c = blockIdx.y * 64 + lane;
    float acc = 0.f;
    for (int k = w; k < n; k += 4) acc += dhr[(int64_t)list[k] * C + c];
    // (the four interleaved partial sums are added in a fixed order: the same bits on every run)
    red[w][lane] = acc;
    __syncthreads();
    if (w == 0) dW[(int64_t)c * V + id] += (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// ================================================================================================ ResBlock head
__global__ __launch_bounds__(128) void relu_taps_kernel(const float* __restrict__ x, const float* __restrict__ ref,
                                                        const float* __restrict__ bias, float bias_scale, float* __restrict__ y,
                                                        float* __restrict__ taps, int L) {
    const int64_t row = blockIdx.x;
    const int l = (int)(row % L), c = 4 * threadIdx.x;
    f32x4 z[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int ll = l + k - 1;
        if (ll >= 0 && ll < L) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (row + k - 1) * C + c);
            const f32x4 rv = *reinterpret_cast<const f32x4*>(ref + (row + k - 1) * C + c);
#pragma unroll
            for (int q = 0; q < 4; ++q) z[k][q] = rv[q] > 0.f ? xv[q] : 0.f;
        } else {
            z[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    f32x4 yo = z[1];
    if (bias) yo += bias_scale * *reinterpret_cast<const f32x4*>(bias + c);
    *reinterpret_cast<f32x4*>(y + row * C + c) = yo;
    float* t = taps + row * (3 * C) + 3 * c;           // columns 3 c .. 3 c + 11: (c, k) pairs in Conv1d's weight order
    *reinterpret_cast<f32x4*>(t) = f32x4{z[0][0], z[1][0], z[2][0], z[0][1]};
    *reinterpret_cast<f32x4*>(t + 4) = f32x4{z[1][1], z[2][1], z[0][2], z[1][2]};
    *reinterpret_cast<f32x4*>(t + 8) = f32x4{z[2][2], z[0][3], z[1][3], z[2][3]};
}

__global__ __launch_bounds__(128) void relu_taps_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ dtaps,
                                                            const float* __restrict__ ref, float* __restrict__ dx, int L) {
    const int64_t row = blockIdx.x;
    const int l = (int)(row % L), c = 4 * threadIdx.x;
    f32x4 acc = *reinterpret_cast<const f32x4*>(dy + row * C + c);
    // z[l] appears as tap k of row l - k + 1: k = 0 -> row l + 1, k = 1 -> row l, k = 2 -> row l - 1
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int ll = l - k + 1;
        if (ll < 0 || ll >= L) continue;
        const float* t = dtaps + (row - k + 1) * (3 * C) + 3 * c;
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(t), t1 = *reinterpret_cast<const f32x4*>(t + 4),
                    t2 = *reinterpret_cast<const f32x4*>(t + 8);
        const float v[12] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3], t2[0], t2[1], t2[2], t2[3]};
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] += v[3 * q + k];
    }
    const f32x4 rv = *reinterpret_cast<const f32x4*>(ref + row * C + c);
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = rv[q] > 0.f ? acc[q] : 0.f;
    *reinterpret_cast<f32x4*>(dx + row * C + c) = acc;
}

// ================================================================================================ (tanh +) LayerNorm with dropouts
// One wave per row, lane l holds columns l + 64 e (coalesced), statistics recomputed from x at every level (see the formulas in
// tests/emul_critic.py cln_f / cln_b).  With a = d gamma, m1 = mean(a), m2 = mean(a n), dt = r (a - m1 - n m2), s = 1 - t^2:
//   bwd2 (U = tangent of x):  W = U mpre s, w1 = mean(W), w2 = mean(W n), core = r (W - w1 - n w2)     [= tangent of n]
//        gdy = gamma core mpost;  ggamma = sum_rows d core
//        Q = sum(W dt) / r,  Pn = -r (W m2 + a w2),  p1 = mean(Pn), p2 = mean(Pn n)
//        gx = mpre s [ r (Pn - p1 - n p2) - Q r^2 n / N - 2 t (U mpre) dt ]                              (last term only with the tanh)
constexpr int LN_MAXE = 16;
struct RowStats { float mu, r; };

template <bool TANH, int EMAX>
__device__ __forceinline__ RowStats ln_row(const float* __restrict__ xr, int E, int lane, float eps, float p_pre, uint64_t seed,
                                           uint32_t site, uint64_t idx0, float (&t)[EMAX], float (&mp)[EMAX]) {
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < EMAX; ++e)
        if (e < E) {
            mp[e] = p_pre > 0.f ? dlsg::drop_scale(seed, site, idx0 + lane + 64 * e, p_pre) : 1.f;
            const float v = xr[lane + 64 * e] * mp[e];
            t[e] = TANH ? tanhf(v) : v;
            sum += t[e];
        }
    const float inv = 1.f / (64.f * E);
    const float mu = dlsg::wave_sum(sum) * inv;
    float sq = 0.f;
#pragma unroll
    for (int e = 0; e < EMAX; ++e)
        if (e < E) sq += (t[e] - mu) * (t[e] - mu);
    RowStats st;
    st.mu = mu;
    st.r = rsqrtf(dlsg::wave_sum(sq) * inv + eps);
    return st;
}

inline int ln_blocks(int rows) { return rows < 2048 ? (rows + 3) / 4 : 512; }

template <bool TANH, int EMAX>
__global__ __launch_bounds__(256) void cln_fwd_kernel(const dlsg_cln_args a) {
    const int lane = threadIdx.x & 63, gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4, g = blockIdx.y;
    const int N = a.N, E = N / 64;
    const float* x = a.x[g]; const float* gamma = a.gamma[g]; const float* beta = a.beta[g];
    float* y = a.y[g];
    const uint64_t seed = seed_of(a.seed, a.seed_ptr);
    float t[EMAX], mp[EMAX];
    for (int row = gw; row < a.rows; row += nw) {
        const uint64_t idx0 = (uint64_t)(a.row0 + row) * N;
        const RowStats st = ln_row<TANH, EMAX>(x + (int64_t)row * N, E, lane, a.eps, a.p_pre, seed, a.site_pre + g, idx0, t, mp);
#pragma unroll
        for (int e = 0; e < EMAX; ++e)
            if (e < E) {
                const int j = lane + 64 * e;
                float v = (t[e] - st.mu) * st.r * gamma[j] + beta[j];
                if (a.p_post > 0.f) v *= dlsg::drop_scale(seed, a.site_post + g, idx0 + j, a.p_post);
                y[(int64_t)row * N + j] = v;
            }
    }
}

template <bool TANH, int EMAX>
__global__ __launch_bounds__(256) void cln_bwd_kernel(const dlsg_cln_args a) {
    const int lane = threadIdx.x & 63, gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4, g = blockIdx.y;
    const int N = a.N, E = N / 64;
    const float* x = a.x[g]; const float* gamma = a.gamma[g];
    float* dx = a.dx[g];
    const bool want = a.dgamma[g] != nullptr || a.defer;
    float* part = a.ws + (int64_t)g * 2 * gridDim.x * N;                    // [group][dgamma | dbeta][workgroup][N]
    const uint64_t seed = seed_of(a.seed, a.seed_ptr);
    float t[EMAX], mp[EMAX], gam[EMAX], pg[EMAX], pb[EMAX];
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
        pg[e] = pb[e] = 0.f;
        gam[e] = e < E ? gamma[lane + 64 * e] : 0.f;
    }
    for (int row = gw; row < a.rows; row += nw) {
        const uint64_t idx0 = (uint64_t)(a.row0 + row) * N;
        const RowStats st = ln_row<TANH, EMAX>(x + (int64_t)row * N, E, lane, a.eps, a.p_pre, seed, a.site_pre + g, idx0, t, mp);
        float av[EMAX], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < EMAX; ++e)
            if (e < E) {
                const int64_t o = (int64_t)row * N + lane + 64 * e;
                float d = a.dy[0][g][o];
                if (a.ndy > 1) d += a.dy[1][g][o];
                if (a.ndy > 2) d += a.dy[2][g][o];
                if (a.p_post > 0.f) d *= dlsg::drop_scale(seed, a.site_post + g, idx0 + lane + 64 * e, a.p_post);
                const float n = (t[e] - st.mu) * st.r;
                av[e] = d * gam[e];
                s1 += av[e];
                s2 += av[e] * n;
                pg[e] += d * n;
                pb[e] += d;
            }
        const float inv = 1.f / (64.f * E);
        const float m1 = dlsg::wave_sum(s1) * inv, m2 = dlsg::wave_sum(s2) * inv;
        const bool acc = row >= a.acc_lo && row < a.acc_hi;
#pragma unroll
        for (int e = 0; e < EMAX; ++e)
            if (e < E) {
                const int64_t o = (int64_t)row * N + lane + 64 * e;
                const float n = (t[e] - st.mu) * st.r;
                float v = st.r * (av[e] - m1 - n * m2);
                if (TANH) v *= 1.f - t[e] * t[e];
                v *= mp[e];
                dx[o] = acc ? dx[o] + v : v;
            }
    }
    if (!want) return;
    __shared__ float red[2][4][64 * EMAX];
    const int w = threadIdx.x >> 6;
#pragma unroll
    for (int e = 0; e < EMAX; ++e)
        if (e < E) {
            red[0][w][lane + 64 * e] = pg[e];
            red[1][w][lane + 64 * e] = pb[e];
        }
    __syncthreads();
    for (int j = threadIdx.x; j < N; j += 256) {
        part[(int64_t)blockIdx.x * N + j] = (red[0][0][j] + red[0][1][j]) + (red[0][2][j] + red[0][3][j]);
        part[((int64_t)gridDim.x + blockIdx.x) * N + j] = (red[1][0][j] + red[1][1][j]) + (red[1][2][j] + red[1][3][j]);
    }
}

template <bool TANH, int EMAX>
__global__ __launch_bounds__(256) void cln_bwd2_kernel(const dlsg_cln_args a) {
    const int lane = threadIdx.x & 63, gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4, g = blockIdx.y;
    const int N = a.N, E = N / 64;
    const float* x = a.x[g]; const float* gamma = a.gamma[g]; const float* U = a.U[g];
    float* gx = a.gx[g]; float* gdy = a.gdy[g];
    float* part = a.ws + (int64_t)g * 2 * gridDim.x * N;
    const uint64_t seed = seed_of(a.seed, a.seed_ptr);
    float t[EMAX], mp[EMAX], gam[EMAX], pgg[EMAX];
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
        pgg[e] = 0.f;
        gam[e] = e < E ? gamma[lane + 64 * e] : 0.f;
    }
    const float inv = 1.f / (64.f * E);
    for (int row = gw; row < a.rows; row += nw) {
        const uint64_t idx0 = (uint64_t)(a.row0 + row) * N;
        const RowStats st = ln_row<TANH, EMAX>(x + (int64_t)row * N, E, lane, a.eps, a.p_pre, seed, a.site_pre + g, idx0, t, mp);
        float d[EMAX], W[EMAX], Uv[EMAX], mq[EMAX], s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
#pragma unroll
        for (int e = 0; e < EMAX; ++e)
            if (e < E) {
                const int64_t o = (int64_t)row * N + lane + 64 * e;
                float dv = a.dy[0][g][o];
                if (a.ndy > 1) dv += a.dy[1][g][o];
                if (a.ndy > 2) dv += a.dy[2][g][o];
                mq[e] = a.p_post > 0.f ? dlsg::drop_scale(seed, a.site_post + g, idx0 + lane + 64 * e, a.p_post) : 1.f;
                d[e] = dv * mq[e];
                const float u = Uv[e] = U[o] * mp[e];
                const float n = (t[e] - st.mu) * st.r, av = d[e] * gam[e];
                W[e] = TANH ? u * (1.f - t[e] * t[e]) : u;
                s1 += av;
                s2 += av * n;
                s3 += W[e];
                s4 += W[e] * n;
            }
        const float m1 = dlsg::wave_sum(s1) * inv, m2 = dlsg::wave_sum(s2) * inv;
        const float w1 = dlsg::wave_sum(s3) * inv, w2 = dlsg::wave_sum(s4) * inv;
        float Pn[EMAX], q = 0.f, s5 = 0.f, s6 = 0.f;
#pragma unroll
        for (int e = 0; e < EMAX; ++e)
            if (e < E) {
                const float n = (t[e] - st.mu) * st.r, av = d[e] * gam[e];
                const float dt = st.r * (av - m1 - n * m2);
                q += W[e] * dt;
                Pn[e] = -st.r * (W[e] * m2 + av * w2);
                s5 += Pn[e];
                s6 += Pn[e] * n;
                const float core = st.r * (W[e] - w1 - n * w2);
                gdy[(int64_t)row * N + lane + 64 * e] = gam[e] * core * mq[e];
                pgg[e] += d[e] * core;
            }
        const float Q = dlsg::wave_sum(q) / st.r, p1 = dlsg::wave_sum(s5) * inv, p2 = dlsg::wave_sum(s6) * inv;
#pragma unroll
        for (int e = 0; e < EMAX; ++e)
            if (e < E) {
                const float n = (t[e] - st.mu) * st.r;
                float G = st.r * (Pn[e] - p1 - n * p2) - Q * st.r * st.r * n * inv;
                if (TANH) {
                    const float s = 1.f - t[e] * t[e];
                    const float av = d[e] * gam[e];
                    const float dt = st.r * (av - m1 - n * m2);
                    G = (G - 2.f * t[e] * Uv[e] * dt) * s;
                }
                gx[(int64_t)row * N + lane + 64 * e] = G * mp[e];
            }
    }
    __shared__ float red[4][64 * EMAX];
    const int w = threadIdx.x >> 6;
#pragma unroll
    for (int e = 0; e < EMAX; ++e)
        if (e < E) red[w][lane + 64 * e] = pgg[e];
    __syncthreads();
    for (int j = threadIdx.x; j < N; j += 256)
        part[(int64_t)blockIdx.x * N + j] = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
}

// out0[g][j] = sum_w part[g][0][w][j] (+ extra[g][0][j]), out1 likewise (K == 2); fixed order.  grid (N / 64, K, groups).
// second-order form (zero1): out1 = 0 (the derivative of dbeta vanishes).
__global__ __launch_bounds__(256) void cln_colsum_kernel(const dlsg_cln_args a, int nw, int K, int mode) {
    __shared__ float red[4][64];
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6, j = blockIdx.x * 64 + c, g = blockIdx.z, k = blockIdx.y, N = a.N;
    float* out;
    if (mode == 0) {
        out = k == 0 ? a.dgamma[g] : a.dbeta[g];
        if (!out) return;
    } else {
        out = a.gpart[g] + k * N;
        if (k == 1) { if (q == 0) out[j] = 0.f; return; }
    }
    const float* p = a.ws + ((int64_t)g * 2 + k) * nw * N;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int w = q;
    for (; w + 28 < nw; w += 32) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] += p[(int64_t)(w + 4 * u) * N + j];
    }
    for (; w < nw; w += 4) acc[0] += p[(int64_t)w * N + j];
    red[q][c] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (q == 0) {
        float r = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
        if (mode == 0 && a.extra[g]) r += a.extra[g][k * N + j];
        out[j] = r;
    }
}

#define CLN_DISPATCH(KERNEL)                                                                                    \
    do {                                                                                                        \
        if (a->N <= 512) {                                                                                      \
            if (a->pre_tanh) hipLaunchKernelGGL((KERNEL<true, 8>), grid, block, 0, ST(stream), *a);             \
            else hipLaunchKernelGGL((KERNEL<false, 8>), grid, block, 0, ST(stream), *a);                        \
        } else {                                                                                                \
            if (a->pre_tanh) hipLaunchKernelGGL((KERNEL<true, 16>), grid, block, 0, ST(stream), *a);            \
            else hipLaunchKernelGGL((KERNEL<false, 16>), grid, block, 0, ST(stream), *a);                       \
        }                                                                                                       \
    } while (0)

bool cln_ok(const dlsg_cln_args* a) {
    return a && a->rows >= 1 && a->N >= 64 && a->N % 64 == 0 && a->N <= 64 * LN_MAXE && a->groups >= 1 && a->groups <= DLSG_CLN_MAXG &&
           a->p_pre >= 0.f && a->p_pre < 1.f && a->p_post >= 0.f && a->p_post < 1.f;
}

// ================================================================================================ masked self-attention core
__device__ __forceinline__ void sa_softmax_rows(float* S, const float* sm, int L) {
    // rows i < L of S (L x L, ld LDW): masked (sm[i] sm[j] <= 0 -> -9e15) softmax over j; one thread per row
    const int i = threadIdx.x;
    if (i < L) {
        float mx = -3.0e38f;
        for (int j = 0; j < L; ++j) {
            const float v = (sm[i] * sm[j] > 0.f) ? S[i * LDW + j] : -9e15f;
            S[i * LDW + j] = v;
            mx = fmaxf(mx, v);
        }
        float sum = 0.f;
        for (int j = 0; j < L; ++j) {
            const float e = __expf(S[i * LDW + j] - mx);
            S[i * LDW + j] = e;
            sum += e;
        }
        const float inv = 1.f / sum;
        for (int j = 0; j < L; ++j) S[i * LDW + j] *= inv;
    }
    __syncthreads();
}

__global__ __launch_bounds__(NT) void sa_fwd_kernel(const dlsg_crit_sa_args a) {
    __shared__ __attribute__((aligned(16))) float S[LMAX * LDW], red[RED], sm[LMAX];
    const int i0 = blockIdx.x, L = a.L;
    const float* K = a.KQV + (int64_t)i0 * L * 3 * C;
    zero_lds(S, LMAX * LDW);
    if (threadIdx.x < L) sm[threadIdx.x] = a.smask[(i0 % a.B) * L + threadIdx.x];
    __syncthreads();
    pairdot(K, 3 * C, L, K + C, 3 * C, L, a.scale, S, LDW, red);
    sa_softmax_rows(S, sm, L);
    for (int k = threadIdx.x; k < L * L; k += NT) a.w[(int64_t)i0 * L * L + k] = S[(k / L) * LDW + (k % L)];
    chanprod<LMAX, false, false>(S, LDW, K + 2 * C, 3 * C, nullptr, nullptr, 0, L, L, 1.f, a.ctx + (int64_t)i0 * L * C, C, false);
}

__global__ __launch_bounds__(NT) void sa_bwd_kernel(const dlsg_crit_sa_args a) {
    __shared__ __attribute__((aligned(16))) float W[LMAX * LDW], D[LMAX * LDW], red[RED], sm[LMAX];
    const int i0 = blockIdx.x, L = a.L;
    const float* K = a.KQV + (int64_t)i0 * L * 3 * C;
    const float* dctx = a.dctx + (int64_t)i0 * L * C;
    float* dK = a.dKQV + (int64_t)i0 * L * 3 * C;
    const bool acc = i0 >= a.acc_lo && i0 < a.acc_hi;
    zero_lds(W, LMAX * LDW); zero_lds(D, LMAX * LDW);
    if (threadIdx.x < L) sm[threadIdx.x] = a.smask[(i0 % a.B) * L + threadIdx.x];
    __syncthreads();
    for (int k = threadIdx.x; k < L * L; k += NT) W[(k / L) * LDW + (k % L)] = a.w[(int64_t)i0 * L * L + k];
    pairdot(dctx, C, L, K + 2 * C, 3 * C, L, 1.f, D, LDW, red);                 // dw = dctx V^T (ends with a barrier)
    chanprod<LMAX, true, false>(W, LDW, dctx, C, nullptr, nullptr, 0, L, L, 1.f, dK + 2 * C, 3 * C, acc);   // dV = w^T dctx
    if (threadIdx.x < L) {
        const int i = threadIdx.x;
        float r = 0.f;
        for (int j = 0; j < L; ++j) r += W[i * LDW + j] * D[i * LDW + j];
        for (int j = 0; j < L; ++j) D[i * LDW + j] = (sm[i] * sm[j] > 0.f) ? W[i * LDW + j] * (D[i * LDW + j] - r) : 0.f;
    }
    __syncthreads();
    chanprod<LMAX, false, false>(D, LDW, K + C, 3 * C, nullptr, nullptr, 0, L, L, a.scale, dK, 3 * C, acc);          // dK = scale dlg Q
    chanprod<LMAX, true, false>(D, LDW, K, 3 * C, nullptr, nullptr, 0, L, L, a.scale, dK + C, 3 * C, acc);          // dQ = scale dlg^T K
}

__global__ __launch_bounds__(NT) void sa_bwd2_kernel(const dlsg_crit_sa_args a) {
    // w, wd (tangent of w), dlg, dlgd
    __shared__ __attribute__((aligned(16))) float W[LMAX * LDW], Wd[LMAX * LDW], D[LMAX * LDW], Dd[LMAX * LDW], red[RED], sm[LMAX];
    const int i0 = blockIdx.x, L = a.L;
    const float* K = a.KQV + (int64_t)i0 * L * 3 * C;
    const float* Ud = a.U + (int64_t)i0 * L * 3 * C;
    const float* dctx = a.dctx + (int64_t)i0 * L * C;
    float* g = a.gKQV + (int64_t)i0 * L * 3 * C;
    zero_lds(W, LMAX * LDW); zero_lds(Wd, LMAX * LDW); zero_lds(D, LMAX * LDW); zero_lds(Dd, LMAX * LDW);
    if (threadIdx.x < L) sm[threadIdx.x] = a.smask[(i0 % a.B) * L + threadIdx.x];
    __syncthreads();
    // w = softmax(scale K Q^T): the forward saved it (a.w) for these captions; Sd = scale (Kd Q^T + K Qd^T) on unmasked entries
    for (int k = threadIdx.x; k < L * L; k += NT) W[(k / L) * LDW + (k % L)] = a.w[(int64_t)i0 * L * L + k];
    pairdot(Ud, 3 * C, L, K + C, 3 * C, L, a.scale, Wd, LDW, red);
    pairdot(K, 3 * C, L, Ud + C, 3 * C, L, a.scale, D, LDW, red);
    if (threadIdx.x < L) {
        const int i = threadIdx.x;
        float r = 0.f;
        for (int j = 0; j < L; ++j) {
            const float sd = (sm[i] * sm[j] > 0.f) ? Wd[i * LDW + j] + D[i * LDW + j] : 0.f;
            Wd[i * LDW + j] = sd;
            r += W[i * LDW + j] * sd;
        }
        for (int j = 0; j < L; ++j) Wd[i * LDW + j] = W[i * LDW + j] * (Wd[i * LDW + j] - r);      // wd
    }
    __syncthreads();
    // tangent of ctx = wd V + w Vd
    chanprod<LMAX, false, true>(Wd, LDW, K + 2 * C, 3 * C, W, Ud + 2 * C, 3 * C, L, L, 1.f, a.Uctx + (int64_t)i0 * L * C, C, false);
    // derivative of dV = wd^T dctx
    chanprod<LMAX, true, false>(Wd, LDW, dctx, C, nullptr, nullptr, 0, L, L, 1.f, g + 2 * C, 3 * C, false);
    // dw = dctx V^T, dwd = dctx Vd^T
    pairdot(dctx, C, L, K + 2 * C, 3 * C, L, 1.f, D, LDW, red);
    pairdot(dctx, C, L, Ud + 2 * C, 3 * C, L, 1.f, Dd, LDW, red);
    if (threadIdx.x < L) {
        const int i = threadIdx.x;
        float r = 0.f, rd = 0.f;
        for (int j = 0; j < L; ++j) {
            r += W[i * LDW + j] * D[i * LDW + j];
            rd += Wd[i * LDW + j] * D[i * LDW + j] + W[i * LDW + j] * Dd[i * LDW + j];
        }
        for (int j = 0; j < L; ++j) {
            const bool on = sm[i] * sm[j] > 0.f;
            const float dw = D[i * LDW + j], dwd = Dd[i * LDW + j], w = W[i * LDW + j], wd = Wd[i * LDW + j];
            D[i * LDW + j] = on ? w * (dw - r) : 0.f;                                   // dlg
            Dd[i * LDW + j] = on ? wd * (dw - r) + w * (dwd - rd) : 0.f;                // dlgd
        }
    }
    __syncthreads();
    chanprod<LMAX, false, true>(Dd, LDW, K + C, 3 * C, D, Ud + C, 3 * C, L, L, a.scale, g, 3 * C, false);           // d(dK) = scale (dlgd Q + dlg Qd)
    chanprod<LMAX, true, true>(Dd, LDW, K, 3 * C, D, Ud, 3 * C, L, L, a.scale, g + C, 3 * C, false);                // d(dQ) = scale (dlgd^T K + dlg^T Kd)
}

// ================================================================================================ word -> proposal graph (PSLScore2)
// softmax over the words (rows l < L) per proposal column t < T of S (L x T, ld LDT), in place
__device__ __forceinline__ void col_softmax(float* S, int L, int T) {
    const int t = threadIdx.x;
    if (t < T) {
        float mx = -3.0e38f;
        for (int l = 0; l < L; ++l) mx = fmaxf(mx, S[l * LDT + t]);
        float sum = 0.f;
        for (int l = 0; l < L; ++l) {
            const float e = __expf(S[l * LDT + t] - mx);
            S[l * LDT + t] = e;
            sum += e;
        }
        const float inv = 1.f / sum;
        for (int l = 0; l < L; ++l) S[l * LDT + t] *= inv;
    }
    __syncthreads();
}

// Y[l][c] (=|+=) sum_t (M1[l][t] G[t][c] + alpha2 M2[l][t] E[t][c]), l < L: two L x T matrices (LDS, zero-padded rows of TMAX)
// against T proposal rows each (global), per channel pair
__device__ __forceinline__ void words_from_proposals(const float* M1, const float* __restrict__ G, const float* M2, float alpha2,
                                                     const float* __restrict__ E, int L, int T, float* __restrict__ Y, bool accum) {
    const int tid = threadIdx.x;
    f32x2 g2[TMAX], e2[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        g2[t] = t < T ? ld2(G + (int64_t)t * C + 2 * tid) : f32x2{0.f, 0.f};
        e2[t] = t < T ? ld2(E + (int64_t)t * C + 2 * tid) : f32x2{0.f, 0.f};
    }
    constexpr int IC = 8;               // (Y += : the old values of IC rows are requested together, see chanprod)
    f32x2 yo[IC];
    for (int l = 0; l < L; ++l) {
        if (accum && (l % IC) == 0) {
#pragma unroll
            for (int u = 0; u < IC; ++u) yo[u] = (l + u < L) ? ld2(Y + (int64_t)(l + u) * C + 2 * tid) : f32x2{0.f, 0.f};
        }
        float m1[TMAX], m2[TMAX];
#pragma unroll
        for (int t = 0; t < TMAX; t += 4) {
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(M1 + l * LDT + t), v2 = *reinterpret_cast<const f32x4*>(M2 + l * LDT + t);
#pragma unroll
            for (int q = 0; q < 4; ++q) { m1[t + q] = v1[q]; m2[t + q] = alpha2 * v2[q]; }
        }
        f32x2 r = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < TMAX; ++t) { r.x += m1[t] * g2[t].x + m2[t] * e2[t].x; r.y += m1[t] * g2[t].y + m2[t] * e2[t].y; }
        float* y = Y + (int64_t)l * C + 2 * tid;
        if (accum) {
            f32x2 o = yo[0];
#pragma unroll
            for (int u = 1; u < IC; ++u) o = (l % IC) == u ? yo[u] : o;
            r.x += o.x; r.y += o.y;
        }
        st2(y, r);
    }
}

__global__ __launch_bounds__(NT) void pattn_fwd_kernel(const dlsg_crit_pattn_args a) {
    __shared__ __attribute__((aligned(16))) float P[LMAX * LDT], adj[LMAX * LDT], red[RED], sm[LMAX];
    const int i0 = blockIdx.x, h = blockIdx.y, L = a.L, T = a.T;
    const float* av = a.a[h] + (int64_t)i0 * L * C;
    const float* ev = a.e[h] + (int64_t)(i0 % a.B) * T * C;
    zero_lds(P, LMAX * LDT); zero_lds(adj, LMAX * LDT);
    if (threadIdx.x < L) sm[threadIdx.x] = a.smask[(i0 % a.B) * L + threadIdx.x];
    __syncthreads();
    pairdot(av, C, L, ev, C, T, a.scale, P, LDT, red);
    col_softmax(P, L, T);
    for (int k = threadIdx.x; k < L * T; k += NT) {
        const int l = k / T, t = k - l * T;
        a.P[h][(int64_t)i0 * L * T + k] = P[l * LDT + t];
        adj[l * LDT + t] = P[l * LDT + t] * sm[l];
    }
    __syncthreads();
    if (threadIdx.x < T) {
        float s = 0.f;
        for (int l = 0; l < L; ++l) s += adj[l * LDT + threadIdx.x];
        a.wgt[h][(int64_t)i0 * T + threadIdx.x] = s;
    }
    chanprod<LMAX, true, false>(adj, LDT, av, C, nullptr, nullptr, 0, T, L, 1.f, a.aggpre[h] + (int64_t)i0 * T * C, C, false);   // adj^T a
}

__global__ __launch_bounds__(NT) void pattn_bwd_kernel(const dlsg_crit_pattn_args a) {
    __shared__ __attribute__((aligned(16))) float P[LMAX * LDT], adj[LMAX * LDT], dS[LMAX * LDT], red[RED], sm[LMAX];
    const int i0 = blockIdx.x, h = blockIdx.y, L = a.L, T = a.T;
    const float* av = a.a[h] + (int64_t)i0 * L * C;
    const float* ev = a.e[h] + (int64_t)(i0 % a.B) * T * C;
    const float* dg = a.d_agg[h] + (int64_t)i0 * T * C;
    const bool acc = i0 >= a.acc_lo && i0 < a.acc_hi;
    zero_lds(P, LMAX * LDT); zero_lds(adj, LMAX * LDT); zero_lds(dS, LMAX * LDT);
    if (threadIdx.x < L) sm[threadIdx.x] = a.smask[(i0 % a.B) * L + threadIdx.x];
    __syncthreads();
    for (int k = threadIdx.x; k < L * T; k += NT) {
        const int l = k / T, t = k - l * T;
        const float p = a.P[h][(int64_t)i0 * L * T + k];
        P[l * LDT + t] = p;
        adj[l * LDT + t] = p * sm[l];
    }
    pairdot(av, C, L, dg, C, T, 1.f, dS, LDT, red);                                // a . d_agg (ends with a barrier)
    if (threadIdx.x < T) {
        const int t = threadIdx.x;
        const float dw = a.d_wgt[h][(int64_t)i0 * T + t];
        float rho = 0.f;
        for (int l = 0; l < L; ++l) {
            const float dP = (dS[l * LDT + t] + dw) * sm[l];
            dS[l * LDT + t] = dP;
            rho += P[l * LDT + t] * dP;
        }
        for (int l = 0; l < L; ++l) dS[l * LDT + t] = P[l * LDT + t] * (dS[l * LDT + t] - rho);
    }
    __syncthreads();
    words_from_proposals(adj, dg, dS, a.scale, ev, L, T, a.da[h] + (int64_t)i0 * L * C, acc);                       // da = adj d_agg + scale dS e
    if (a.de[h]) chanprod<LMAX, true, false>(dS, LDT, av, C, nullptr, nullptr, 0, T, L, a.scale, a.de[h] + (int64_t)i0 * T * C, C, false);
}

__global__ __launch_bounds__(NT) void pattn_bwd2_kernel(const dlsg_crit_pattn_args a) {
    __shared__ __attribute__((aligned(16))) float P[LMAX * LDT], Pd[LMAX * LDT], dS[LMAX * LDT], dSd[LMAX * LDT], adj[LMAX * LDT],
        adjd[LMAX * LDT], red[RED], sm[LMAX];
    const int i0 = blockIdx.x, h = blockIdx.y, L = a.L, T = a.T, tid = threadIdx.x;
    const float* av = a.a[h] + (int64_t)i0 * L * C;
    const float* ad = a.Ua[h] + (int64_t)i0 * L * C;
    const float* ev = a.e[h] + (int64_t)(i0 % a.B) * T * C;
    const float* dg = a.d_agg[h] + (int64_t)i0 * T * C;
    zero_lds(P, LMAX * LDT); zero_lds(Pd, LMAX * LDT); zero_lds(dS, LMAX * LDT); zero_lds(dSd, LMAX * LDT);
    zero_lds(adj, LMAX * LDT); zero_lds(adjd, LMAX * LDT);
    if (tid < L) sm[tid] = a.smask[(i0 % a.B) * L + tid];
    __syncthreads();
    for (int k = tid; k < L * T; k += NT) P[(k / T) * LDT + (k % T)] = a.P[h][(int64_t)i0 * L * T + k];      // saved by the forward
    pairdot(ad, C, L, ev, C, T, a.scale, Pd, LDT, red);                            // Sd
    pairdot(av, C, L, dg, C, T, 1.f, dS, LDT, red);                                // a . d_agg
    pairdot(ad, C, L, dg, C, T, 1.f, dSd, LDT, red);                               // ad . d_agg
    if (tid < T) {
        const int t = tid;
        const float dw = a.d_wgt[h][(int64_t)i0 * T + t];
        float pi = 0.f;
        for (int l = 0; l < L; ++l) pi += P[l * LDT + t] * Pd[l * LDT + t];
        float rho = 0.f, rhod = 0.f, wd = 0.f;
        for (int l = 0; l < L; ++l) {
            const float p = P[l * LDT + t], pd = p * (Pd[l * LDT + t] - pi);
            Pd[l * LDT + t] = pd;
            adj[l * LDT + t] = p * sm[l];
            adjd[l * LDT + t] = pd * sm[l];
            wd += pd * sm[l];
            const float dP = (dS[l * LDT + t] + dw) * sm[l], dPd = dSd[l * LDT + t] * sm[l];
            dS[l * LDT + t] = dP; dSd[l * LDT + t] = dPd;
            rho += p * dP;
            rhod += pd * dP + p * dPd;
        }
        a.Uwgt[h][(int64_t)i0 * T + t] = wd;
        for (int l = 0; l < L; ++l) {
            const float p = P[l * LDT + t], pd = Pd[l * LDT + t], dP = dS[l * LDT + t], dPd = dSd[l * LDT + t];
            dS[l * LDT + t] = p * (dP - rho);
            dSd[l * LDT + t] = pd * (dP - rho) + p * (dPd - rhod);
        }
    }
    __syncthreads();
    // tangent of aggpre = adjd^T a + adj^T ad
    chanprod<LMAX, true, true>(adjd, LDT, av, C, adj, ad, C, T, L, 1.f, a.Uagg[h] + (int64_t)i0 * T * C, C, false);
    // derivative of de = scale (dSd^T a + dS^T ad)
    chanprod<LMAX, true, true>(dSd, LDT, av, C, dS, ad, C, T, L, a.scale, a.ge[h] + (int64_t)i0 * T * C, C, false);
    // derivative of da = adjd d_agg + scale dSd e
    words_from_proposals(adjd, dg, dSd, a.scale, ev, L, T, a.ga[h] + (int64_t)i0 * L * C, false);
}

// ================================================================================================ text summary + fusion weights
struct LnRow { float mu, r; };

// forward of one caption up to fus; keeps t = tanh(u), n (normalised), mask, sent in registers (channel pair) for the callers
struct TsumFwd {
    f32x2 u, t, n, mp, sent;
    float r;
    float fus[2], fl[2];
};
__device__ __forceinline__ TsumFwd tsum_forward(const dlsg_crit_tsum_args& a, int i0, const float* __restrict__ words, float* adj_s,
                                               float* scratch, float* red) {
    const int tid = threadIdx.x, L = a.L;
    const f32x2 th = ld2(a.theta + 2 * tid);
    rowdots(words, C, L, th, adj_s, scratch);                                 // lg
    if (tid == 0) {
        float mx = -3.0e38f, sum = 0.f;
        for (int l = 0; l < L; ++l) mx = fmaxf(mx, adj_s[l]);
        for (int l = 0; l < L; ++l) { adj_s[l] = __expf(adj_s[l] - mx); sum += adj_s[l]; }
        const float inv = 1.f / sum;
        for (int l = 0; l < L; ++l) adj_s[l] *= inv;
    }
    __syncthreads();
    TsumFwd f;
    f.u = f32x2{0.f, 0.f};
    for (int l = 0; l < L; ++l) {
        const f32x2 wv = ld2(words + (int64_t)l * C + 2 * tid);
        f.u.x += adj_s[l] * wv.x; f.u.y += adj_s[l] * wv.y;
    }
    f.t = f32x2{tanhf(f.u.x), tanhf(f.u.y)};
    const float mu = bsum(f.t.x + f.t.y, red) * (1.f / C);
    const float var = bsum((f.t.x - mu) * (f.t.x - mu) + (f.t.y - mu) * (f.t.y - mu), red) * (1.f / C);
    f.r = rsqrtf(var + a.eps);
    f.n = f32x2{(f.t.x - mu) * f.r, (f.t.y - mu) * f.r};
    const uint64_t seed = seed_of(a.seed, a.seed_ptr);
    const uint64_t idx = (uint64_t)(a.row0 + i0) * C + 2 * tid;
    f.mp = a.p > 0.f ? f32x2{dlsg::drop_scale(seed, a.site, idx, a.p), dlsg::drop_scale(seed, a.site, idx + 1, a.p)} : f32x2{1.f, 1.f};
    const f32x2 ga = ld2(a.gamma + 2 * tid), be = ld2(a.beta + 2 * tid);
    f.sent = f32x2{(f.n.x * ga.x + be.x) * f.mp.x, (f.n.y * ga.y + be.y) * f.mp.y};
    f.fl[0] = bsum(dot2(f.sent, ld2(a.fusion + 2 * tid)), red);
    f.fl[1] = bsum(dot2(f.sent, ld2(a.fusion + C + 2 * tid)), red);
    const float mx = fmaxf(f.fl[0], f.fl[1]);
    const float e0 = __expf(f.fl[0] - mx), e1 = __expf(f.fl[1] - mx);
    f.fus[0] = e0 / (e0 + e1); f.fus[1] = e1 / (e0 + e1);
    return f;
}

__global__ __launch_bounds__(NT) void tsum_fwd_kernel(const dlsg_crit_tsum_args a) {
    __shared__ float adj_s[LMAX], scratch[4 * LMAX], red[4];
    const int i0 = blockIdx.x, tid = threadIdx.x, L = a.L;
    const TsumFwd f = tsum_forward(a, i0, a.words + (int64_t)i0 * L * C, adj_s, scratch, red);
    if (tid < L) a.adj[(int64_t)i0 * L + tid] = adj_s[tid];
    st2(a.u + (int64_t)i0 * C + 2 * tid, f.u);
    st2(a.sent + (int64_t)i0 * C + 2 * tid, f.sent);
    if (tid < 2) a.fus[(int64_t)i0 * 2 + tid] = f.fus[tid];
}

__global__ __launch_bounds__(NT) void tsum_bwd_kernel(const dlsg_crit_tsum_args a) {
    __shared__ float adj_s[LMAX], dlg_s[LMAX], scratch[4 * LMAX], red[4];
    const int i0 = blockIdx.x, tid = threadIdx.x, L = a.L;
    const float* words = a.words + (int64_t)i0 * L * C;
    const bool acc = i0 >= a.acc_lo && i0 < a.acc_hi;
    const TsumFwd f = tsum_forward(a, i0, words, adj_s, scratch, red);
    const float df0 = a.d_fus[(int64_t)i0 * 2], df1 = a.d_fus[(int64_t)i0 * 2 + 1];
    const float q = f.fus[0] * df0 + f.fus[1] * df1;
    const float dfl0 = f.fus[0] * (df0 - q), dfl1 = f.fus[1] * (df1 - q);
    const f32x2 F0 = ld2(a.fusion + 2 * tid), F1 = ld2(a.fusion + C + 2 * tid), ga = ld2(a.gamma + 2 * tid);
    const f32x2 dsent = {dfl0 * F0.x + dfl1 * F1.x, dfl0 * F0.y + dfl1 * F1.y};
    const f32x2 d = {dsent.x * f.mp.x, dsent.y * f.mp.y};
    const f32x2 av = {d.x * ga.x, d.y * ga.y};
    const float m1 = bsum(av.x + av.y, red) * (1.f / C);
    const float m2 = bsum(av.x * f.n.x + av.y * f.n.y, red) * (1.f / C);
    const f32x2 dt = {f.r * (av.x - m1 - f.n.x * m2), f.r * (av.y - m1 - f.n.y * m2)};
    const f32x2 du = {dt.x * (1.f - f.t.x * f.t.x), dt.y * (1.f - f.t.y * f.t.y)};
    rowdots(words, C, L, du, dlg_s, scratch);                                 // dadj
    if (tid == 0) {
        float p = 0.f;
        for (int l = 0; l < L; ++l) p += adj_s[l] * dlg_s[l];
        for (int l = 0; l < L; ++l) dlg_s[l] = adj_s[l] * (dlg_s[l] - p);
    }
    __syncthreads();
    const f32x2 th = ld2(a.theta + 2 * tid);
    f32x2 dth = {0.f, 0.f};
    float* dw = a.dwords + (int64_t)i0 * L * C;
    constexpr int IC = 8;               // rows whose loads (the words; with acc the old dwords) are requested together
    f32x2 yo[IC], wv8[IC];
    for (int l = 0; l < L; ++l) {
        if ((l % IC) == 0) {
#pragma unroll
            for (int u = 0; u < IC; ++u) {
                const bool in = l + u < L;
                wv8[u] = in ? ld2(words + (int64_t)(l + u) * C + 2 * tid) : f32x2{0.f, 0.f};
                yo[u] = (in && acc) ? ld2(dw + (int64_t)(l + u) * C + 2 * tid) : f32x2{0.f, 0.f};
            }
        }
        f32x2 o = yo[0], wv = wv8[0];
#pragma unroll
        for (int u = 1; u < IC; ++u) { o = (l % IC) == u ? yo[u] : o; wv = (l % IC) == u ? wv8[u] : wv; }
        const f32x2 r = {adj_s[l] * du.x + dlg_s[l] * th.x + o.x, adj_s[l] * du.y + dlg_s[l] * th.y + o.y};
        st2(dw + (int64_t)l * C + 2 * tid, r);
        dth.x += dlg_s[l] * wv.x; dth.y += dlg_s[l] * wv.y;
    }
    if (a.part) {
        float* p = a.part + (int64_t)i0 * 5 * C + 2 * tid;
        st2(p, dth);
        st2(p + C, f32x2{d.x * f.n.x, d.y * f.n.y});
        st2(p + 2 * C, d);
        st2(p + 3 * C, f32x2{dfl0 * f.sent.x, dfl0 * f.sent.y});
        st2(p + 4 * C, f32x2{dfl1 * f.sent.x, dfl1 * f.sent.y});
    }
}

__global__ __launch_bounds__(NT) void tsum_bwd2_kernel(const dlsg_crit_tsum_args a) {
    __shared__ float adj_s[LMAX], adjd_s[LMAX], dadj_s[LMAX], dadjd_s[LMAX], tmp_s[LMAX], scratch[4 * LMAX], red[4];
    const int i0 = blockIdx.x, tid = threadIdx.x, L = a.L;
    const float* words = a.words + (int64_t)i0 * L * C;
    const float* Wd = a.U + (int64_t)i0 * L * C;
    const TsumFwd f = tsum_forward(a, i0, words, adj_s, scratch, red);
    const f32x2 th = ld2(a.theta + 2 * tid), ga = ld2(a.gamma + 2 * tid);
    const f32x2 F0 = ld2(a.fusion + 2 * tid), F1 = ld2(a.fusion + C + 2 * tid);
    // ---- tangent of the forward
    rowdots(Wd, C, L, th, adjd_s, scratch);                                   // lgd
    if (tid == 0) {
        float s = 0.f;
        for (int l = 0; l < L; ++l) s += adj_s[l] * adjd_s[l];
        for (int l = 0; l < L; ++l) adjd_s[l] = adj_s[l] * (adjd_s[l] - s);
    }
    __syncthreads();
    f32x2 ud = {0.f, 0.f};
    for (int l = 0; l < L; ++l) {
        const f32x2 wv = ld2(words + (int64_t)l * C + 2 * tid), wd = ld2(Wd + (int64_t)l * C + 2 * tid);
        ud.x += adjd_s[l] * wv.x + adj_s[l] * wd.x; ud.y += adjd_s[l] * wv.y + adj_s[l] * wd.y;
    }
    const f32x2 s_ = {1.f - f.t.x * f.t.x, 1.f - f.t.y * f.t.y};
    const f32x2 td = {s_.x * ud.x, s_.y * ud.y};
    const float w1 = bsum(td.x + td.y, red) * (1.f / C);
    const float w2 = bsum(td.x * f.n.x + td.y * f.n.y, red) * (1.f / C);
    const f32x2 nd = {f.r * (td.x - w1 - f.n.x * w2), f.r * (td.y - w1 - f.n.y * w2)};
    const float rd = -f.r * f.r * w2;
    const f32x2 sentd = {nd.x * ga.x * f.mp.x, nd.y * ga.y * f.mp.y};
    const float fld0 = bsum(dot2(sentd, F0), red), fld1 = bsum(dot2(sentd, F1), red);
    const float sf = f.fus[0] * fld0 + f.fus[1] * fld1;
    const float fusd0 = f.fus[0] * (fld0 - sf), fusd1 = f.fus[1] * (fld1 - sf);
    if (tid == 0) { a.Ufus[(int64_t)i0 * 2] = fusd0; a.Ufus[(int64_t)i0 * 2 + 1] = fusd1; }
    // ---- the backward and its derivative
    const float df0 = a.d_fus[(int64_t)i0 * 2], df1 = a.d_fus[(int64_t)i0 * 2 + 1];
    const float q = f.fus[0] * df0 + f.fus[1] * df1, qd = fusd0 * df0 + fusd1 * df1;
    const float dfl0 = f.fus[0] * (df0 - q), dfl1 = f.fus[1] * (df1 - q);
    const float dfld0 = fusd0 * (df0 - q) - f.fus[0] * qd, dfld1 = fusd1 * (df1 - q) - f.fus[1] * qd;
    const f32x2 d = {(dfl0 * F0.x + dfl1 * F1.x) * f.mp.x, (dfl0 * F0.y + dfl1 * F1.y) * f.mp.y};
    const f32x2 dd = {(dfld0 * F0.x + dfld1 * F1.x) * f.mp.x, (dfld0 * F0.y + dfld1 * F1.y) * f.mp.y};
    const f32x2 av = {d.x * ga.x, d.y * ga.y}, avd = {dd.x * ga.x, dd.y * ga.y};
    const float m1 = bsum(av.x + av.y, red) * (1.f / C), m1d = bsum(avd.x + avd.y, red) * (1.f / C);
    const float m2 = bsum(av.x * f.n.x + av.y * f.n.y, red) * (1.f / C);
    const float m2d = bsum(avd.x * f.n.x + avd.y * f.n.y + av.x * nd.x + av.y * nd.y, red) * (1.f / C);
    const f32x2 base = {av.x - m1 - f.n.x * m2, av.y - m1 - f.n.y * m2};
    const f32x2 dt = {f.r * base.x, f.r * base.y};
    const f32x2 dtd = {rd * base.x + f.r * (avd.x - m1d - nd.x * m2 - f.n.x * m2d), rd * base.y + f.r * (avd.y - m1d - nd.y * m2 - f.n.y * m2d)};
    const f32x2 du = {dt.x * s_.x, dt.y * s_.y};
    const f32x2 dud = {dtd.x * s_.x - 2.f * dt.x * f.t.x * td.x, dtd.y * s_.y - 2.f * dt.y * f.t.y * td.y};
    rowdots(words, C, L, du, dadj_s, scratch);                                // dadj = words . du
    rowdots(Wd, C, L, du, dadjd_s, scratch);                                  // Wd . du
    rowdots(words, C, L, dud, tmp_s, scratch);                                // words . dud
    if (tid == 0) {
        float p = 0.f, pd = 0.f;
        for (int l = 0; l < L; ++l) {
            dadjd_s[l] += tmp_s[l];
            p += adj_s[l] * dadj_s[l];
            pd += adjd_s[l] * dadj_s[l] + adj_s[l] * dadjd_s[l];
        }
        for (int l = 0; l < L; ++l) {
            const float dlg = adj_s[l] * (dadj_s[l] - p);
            const float dlgd = adjd_s[l] * (dadj_s[l] - p) + adj_s[l] * (dadjd_s[l] - pd);
            dadj_s[l] = dlg; dadjd_s[l] = dlgd;
        }
    }
    __syncthreads();
    f32x2 dthd = {0.f, 0.f};
    float* gw = a.gwords + (int64_t)i0 * L * C;
    for (int l = 0; l < L; ++l) {
        st2(gw + (int64_t)l * C + 2 * tid, f32x2{adjd_s[l] * du.x + adj_s[l] * dud.x + dadjd_s[l] * th.x,
                                                adjd_s[l] * du.y + adj_s[l] * dud.y + dadjd_s[l] * th.y});
        const f32x2 wv = ld2(words + (int64_t)l * C + 2 * tid), wd = ld2(Wd + (int64_t)l * C + 2 * tid);
        dthd.x += dadjd_s[l] * wv.x + dadj_s[l] * wd.x; dthd.y += dadjd_s[l] * wv.y + dadj_s[l] * wd.y;
    }
    float* p = a.gpart + (int64_t)i0 * 5 * C + 2 * tid;
    st2(p, dthd);
    st2(p + C, f32x2{dd.x * f.n.x + d.x * nd.x, dd.y * f.n.y + d.y * nd.y});
    st2(p + 2 * C, dd);
    st2(p + 3 * C, f32x2{dfld0 * f.sent.x + dfl0 * sentd.x, dfld0 * f.sent.y + dfl0 * sentd.y});
    st2(p + 4 * C, f32x2{dfld1 * f.sent.x + dfl1 * sentd.x, dfld1 * f.sent.y + dfl1 * sentd.y});
}

// ================================================================================================ pair scores -> critic output
// phase 1, grid (n, 2): pair, score of one (caption, head)
__global__ __launch_bounds__(NT) void score_pair_kernel(const dlsg_crit_score_args a) {
    __shared__ float pr[TMAX], red[4];
    const int i0 = blockIdx.x, h = blockIdx.y, T = a.T, tid = threadIdx.x;
    const float* v = a.v[h] + (int64_t)(i0 % a.B) * T * C;
    const float* s = a.s[h] + (int64_t)i0 * T * C;
    const f32x2 wc = ld2(a.wc[h] + 2 * tid);
    for (int t = 0; t < T; ++t) {
        const f32x2 vv = ld2(v + (int64_t)t * C + 2 * tid), sv = ld2(s + (int64_t)t * C + 2 * tid);
        const float p = bsum(vv.x * sv.x * wc.x + vv.y * sv.y * wc.y, red) + a.bc[h][0];
        if (tid == 0) pr[t] = p;
    }
    __syncthreads();
    if (tid == 0) {
        float num = 0.f, den = 0.f;
        for (int t = 0; t < T; ++t) {
            const float w = a.wgt[h][(int64_t)i0 * T + t];
            a.pair[h][(int64_t)i0 * T + t] = pr[t];
            num += pr[t] * w; den += w;
        }
        a.score[h][i0] = num / den;
    }
}
// phase 2, one workgroup: both[g][h] = mean_b score[h][g B + b]; out[i] = sum_h both[g(i)][h] fus[i][h]
__global__ __launch_bounds__(NT) void score_out_kernel(const dlsg_crit_score_args a) {
    __shared__ float both[8 * 2], red[4];
    const int tid = threadIdx.x, B = a.B;
    for (int k = 0; k < 2 * a.ng; ++k) {
        const int g = k >> 1, h = k & 1;
        float s = 0.f;
        for (int b = tid; b < B; b += NT) s += a.score[h][g * B + b];
        s = bsum(s, red);
        if (tid == 0) { both[k] = s / B; a.both[k] = s / B; }
    }
    __syncthreads();
    for (int i = tid; i < a.n; i += NT) {
        const int g = i / B;
        a.out[i] = both[2 * g] * a.fus[2 * i] + both[2 * g + 1] * a.fus[2 * i + 1];
    }
}

// bwd phase 1, one workgroup: d_both[g][h] = sum_{i in g} d_out[i] fus[i][h] -> scratch[2 g + h]; d_fus; dbc
__global__ __launch_bounds__(NT) void score_bwd_head_kernel(const dlsg_crit_score_args a) {
    __shared__ float red[4];
    const int tid = threadIdx.x, B = a.B;
    float dbc0 = 0.f, dbc1 = 0.f;
    for (int k = 0; k < 2 * a.ng; ++k) {
        const int g = k >> 1, h = k & 1;
        float s = 0.f;
        for (int b = tid; b < B; b += NT) s += a.d_out[g * B + b] * a.fus[2 * (g * B + b) + h];
        s = bsum(s, red);
        if (tid == 0) a.scratch[k] = s;
        if (h == 0) dbc0 += s; else dbc1 += s;
    }
    if (tid == 0 && a.dbc) { a.dbc[0] = dbc0; a.dbc[1] = dbc1; }
    for (int i = tid; i < a.n; i += NT) {
        const int g = i / B;
        const bool acc = i >= a.acc_lo && i < a.acc_hi;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float v = a.d_out[i] * a.both[2 * g + h];
            a.d_fus[2 * i + h] = acc ? a.d_fus[2 * i + h] + v : v;
        }
    }
}
// bwd phase 2, grid (n, 2)
__global__ __launch_bounds__(NT) void score_bwd_kernel(const dlsg_crit_score_args a) {
    __shared__ float dp[TMAX];
    const int i0 = blockIdx.x, h = blockIdx.y, T = a.T, tid = threadIdx.x, g = i0 / a.B;
    const bool acc = i0 >= a.acc_lo && i0 < a.acc_hi;
    if (tid == 0) {
        const float d_score = a.scratch[2 * g + h] / a.B, score = a.score[h][i0];
        float wsum = 0.f;
        for (int t = 0; t < T; ++t) wsum += a.wgt[h][(int64_t)i0 * T + t];
        for (int t = 0; t < T; ++t) {
            const float w = a.wgt[h][(int64_t)i0 * T + t];
            dp[t] = d_score * w / wsum;
            const float dw = d_score * (a.pair[h][(int64_t)i0 * T + t] - score) / wsum;
            float* o = a.d_wgt[h] + (int64_t)i0 * T + t;
            *o = acc ? *o + dw : dw;
        }
    }
    __syncthreads();
    const float* v = a.v[h] + (int64_t)(i0 % a.B) * T * C;
    const float* s = a.s[h] + (int64_t)i0 * T * C;
    const f32x2 wc = ld2(a.wc[h] + 2 * tid);
    f32x2 pw = {0.f, 0.f};
    for (int t = 0; t < T; ++t) {
        const f32x2 vv = ld2(v + (int64_t)t * C + 2 * tid), sv = ld2(s + (int64_t)t * C + 2 * tid);
        f32x2 cs = {dp[t] * vv.x * wc.x * (1.f - sv.x * sv.x), dp[t] * vv.y * wc.y * (1.f - sv.y * sv.y)};
        float* o = a.c_spre[h] + ((int64_t)i0 * T + t) * C + 2 * tid;
        if (acc) { const f32x2 p = ld2(o); cs.x += p.x; cs.y += p.y; }
        st2(o, cs);
        if (a.c_vpre[h])
            st2(a.c_vpre[h] + ((int64_t)i0 * T + t) * C + 2 * tid,
                f32x2{dp[t] * sv.x * wc.x * (1.f - vv.x * vv.x), dp[t] * sv.y * wc.y * (1.f - vv.y * vv.y)});
        pw.x += dp[t] * vv.x * sv.x; pw.y += dp[t] * vv.y * sv.y;
    }
    if (a.part_wc[h]) st2(a.part_wc[h] + (int64_t)i0 * C + 2 * tid, pw);
}

// bwd2 phase 1, grid (n, 2): paird, scored of one (caption, head) -> scratch: [16 + 2 i + h] = scored; the kernel also keeps paird
// in scratch2 = a.scratch + 16 + 2 n ... (T per (i, h))
__global__ __launch_bounds__(NT) void score_bwd2_a_kernel(const dlsg_crit_score_args a) {
    __shared__ float prd[TMAX], red[4];
    const int i0 = blockIdx.x, h = blockIdx.y, T = a.T, tid = threadIdx.x, n = a.n;
    const float* v = a.v[h] + (int64_t)(i0 % a.B) * T * C;
    const float* s = a.s[h] + (int64_t)i0 * T * C;
    const float* us = a.Uspre[h] + (int64_t)i0 * T * C;
    const f32x2 wc = ld2(a.wc[h] + 2 * tid);
    for (int t = 0; t < T; ++t) {
        const f32x2 vv = ld2(v + (int64_t)t * C + 2 * tid), sv = ld2(s + (int64_t)t * C + 2 * tid), uv = ld2(us + (int64_t)t * C + 2 * tid);
        const float p = bsum(vv.x * (1.f - sv.x * sv.x) * uv.x * wc.x + vv.y * (1.f - sv.y * sv.y) * uv.y * wc.y, red);
        if (tid == 0) prd[t] = p;
    }
    __syncthreads();
    if (tid == 0) {
        float wsum = 0.f, wsumd = 0.f, numd = 0.f;
        for (int t = 0; t < T; ++t) {
            const float w = a.wgt[h][(int64_t)i0 * T + t], wd = a.Uwgt[h][(int64_t)i0 * T + t];
            wsum += w; wsumd += wd;
            numd += prd[t] * w + a.pair[h][(int64_t)i0 * T + t] * wd;
            a.scratch[16 + 2 * n + ((int64_t)h * n + i0) * TMAX + t] = prd[t];
        }
        a.scratch[16 + 2 * i0 + h] = (numd - a.score[h][i0] * wsumd) / wsum;
    }
}
// bwd2 phase 2, one workgroup: bothd[h] = mean_b scored; d_both[h], d_bothd[h]; g_fus; g_dbc
__global__ __launch_bounds__(NT) void score_bwd2_b_kernel(const dlsg_crit_score_args a) {
    __shared__ float red[4];
    const int tid = threadIdx.x, n = a.n;
    float r[6];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        for (int i = tid; i < n; i += NT) {
            s0 += a.scratch[16 + 2 * i + h];
            s1 += a.d_out[i] * a.fus[2 * i + h];
            s2 += a.d_out[i] * a.Ufus[2 * i + h];
        }
        r[h] = bsum(s0, red) / n;               // bothd
        r[2 + h] = bsum(s1, red);               // d_both
        r[4 + h] = bsum(s2, red);               // d_bothd
    }
    if (tid < 6) a.scratch[tid] = r[tid];
    if (tid == 0) { a.dbc[0] = r[4]; a.dbc[1] = r[5]; }
    for (int i = tid; i < n; i += NT) {
        a.d_fus[2 * i] = a.d_out[i] * r[0];
        a.d_fus[2 * i + 1] = a.d_out[i] * r[1];
    }
}
// bwd2 phase 3, grid (n, 2): the derivatives of bwd's per-caption outputs
__global__ __launch_bounds__(NT) void score_bwd2_c_kernel(const dlsg_crit_score_args a) {
    __shared__ float dp[TMAX], dpd[TMAX];
    const int i0 = blockIdx.x, h = blockIdx.y, T = a.T, tid = threadIdx.x, n = a.n;
    if (tid == 0) {
        const float d_score = a.scratch[2 + h] / n, d_scored = a.scratch[4 + h] / n;
        const float score = a.score[h][i0], scored = a.scratch[16 + 2 * i0 + h];
        float wsum = 0.f, wsumd = 0.f;
        for (int t = 0; t < T; ++t) { wsum += a.wgt[h][(int64_t)i0 * T + t]; wsumd += a.Uwgt[h][(int64_t)i0 * T + t]; }
        for (int t = 0; t < T; ++t) {
            const float w = a.wgt[h][(int64_t)i0 * T + t], wd = a.Uwgt[h][(int64_t)i0 * T + t];
            const float pr = a.pair[h][(int64_t)i0 * T + t], prd = a.scratch[16 + 2 * n + ((int64_t)h * n + i0) * TMAX + t];
            dp[t] = d_score * w / wsum;
            dpd[t] = d_scored * w / wsum + d_score * (wd / wsum - w * wsumd / (wsum * wsum));
            a.d_wgt[h][(int64_t)i0 * T + t] = d_scored * (pr - score) / wsum +
                                              d_score * ((prd - scored) / wsum - (pr - score) * wsumd / (wsum * wsum));
        }
    }
    __syncthreads();
    const float* v = a.v[h] + (int64_t)(i0 % a.B) * T * C;
    const float* s = a.s[h] + (int64_t)i0 * T * C;
    const float* us = a.Uspre[h] + (int64_t)i0 * T * C;
    const f32x2 wc = ld2(a.wc[h] + 2 * tid);
    f32x2 pw = {0.f, 0.f};
    for (int t = 0; t < T; ++t) {
        const f32x2 vv = ld2(v + (int64_t)t * C + 2 * tid), sv = ld2(s + (int64_t)t * C + 2 * tid), uv = ld2(us + (int64_t)t * C + 2 * tid);
        const f32x2 q = {1.f - sv.x * sv.x, 1.f - sv.y * sv.y};
        const f32x2 sd = {q.x * uv.x, q.y * uv.y};
        st2(a.c_spre[h] + ((int64_t)i0 * T + t) * C + 2 * tid,
            f32x2{vv.x * wc.x * (dpd[t] * q.x - 2.f * dp[t] * sv.x * sd.x), vv.y * wc.y * (dpd[t] * q.y - 2.f * dp[t] * sv.y * sd.y)});
        st2(a.c_vpre[h] + ((int64_t)i0 * T + t) * C + 2 * tid,
            f32x2{(dpd[t] * sv.x + dp[t] * sd.x) * wc.x * (1.f - vv.x * vv.x), (dpd[t] * sv.y + dp[t] * sd.y) * wc.y * (1.f - vv.y * vv.y)});
        pw.x += vv.x * (dpd[t] * sv.x + dp[t] * sd.x); pw.y += vv.y * (dpd[t] * sv.y + dp[t] * sd.y);
    }
    st2(a.part_wc[h] + (int64_t)i0 * C + 2 * tid, pw);
}

// ================================================================================================ gradient penalty
__global__ __launch_bounds__(NT) void gp_q_kernel(const float* __restrict__ g, const float* __restrict__ gG, float* __restrict__ q, int L) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    float s = 0.f;
    for (int l = 0; l < L; ++l) s += dot2(ld2(g + ((int64_t)b * L + l) * C + 2 * tid), ld2(gG + ((int64_t)b * L + l) * C + 2 * tid));
    s = bsum(s, red);
    if (tid == 0) q[b] = s;
}
__global__ __launch_bounds__(NT) void gp_out_kernel(const float* __restrict__ g, const float* __restrict__ gG, const float* __restrict__ out,
                                                    const float* __restrict__ q, float* __restrict__ stats, float* __restrict__ vseed,
                                                    float* __restrict__ gsc, int B, int L) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float qb = q[b], gn = sqrtf(fmaxf(qb, 1e-24f));
    const float c = qb > 1e-24f ? (gn - 1.f) / (B * gn) : 0.f;
    for (int l = 0; l < L; ++l) {
        const int64_t o = ((int64_t)b * L + l) * C + 2 * tid;
        const f32x2 gv = ld2(g + o), Gv = ld2(gG + o);
        st2(vseed + o, f32x2{20.f * c * Gv.x, 20.f * c * Gv.y});
        st2(gsc + o, f32x2{10.f * c * gv.x, 10.f * c * gv.y});
    }
    if (b == 0) {
        float r = 0.f, f = 0.f, p = 0.f;
        for (int i = tid; i < B; i += NT) {
            r += out[i]; f += out[B + i];
            const float gi = sqrtf(fmaxf(q[i], 1e-24f));
            p += (gi - 1.f) * (gi - 1.f);
        }
        r = bsum(r, red) / B; f = bsum(f, red) / B; p = bsum(p, red) / B;
        if (tid == 0) { stats[0] = f - r + 10.f * p; stats[1] = r; stats[2] = f; stats[3] = p; stats[4] = r - f; }
    }
}

// ================================================================================================ top-k proposals / inverse gather
__global__ __launch_bounds__(64) void topk_kernel(const float* __restrict__ alpha, int64_t sa, int64_t lda, int na,
                                                  const float* __restrict__ smask, int64_t* __restrict__ idx, int B, int L, int P, int T) {
    const int b = blockIdx.x, h = blockIdx.y, p = threadIdx.x;
    __shared__ float sums[64];
    float s = -3.0e38f;
    if (p < P) {
        s = 0.f;
        const int col = (h == 0 ? 0 : na - P) + p;
        for (int l = 0; l < L; ++l) s += alpha[(int64_t)b * sa + (int64_t)l * lda + col] * smask[b * L + l];
    }
    sums[p] = s;
    __syncthreads();
    if (p == 0) {
        for (int t = 0; t < T; ++t) {
            int best = 0;
            for (int k = 1; k < P; ++k)
                if (sums[k] > sums[best]) best = k;
            idx[((int64_t)h * B + b) * T + t] = ((int64_t)h * B + b) * P + best;
            sums[best] = -3.0e38f;
        }
    }
}
// dst rows (R / per groups of `per` rows, T selected per group): row r of dst is src[j] if idx[j] == r for the j of its group
__global__ __launch_bounds__(128) void unselect_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx,
                                                       float* __restrict__ dst, int T, int per, int n) {
    const int r = blockIdx.x, grp = r / per;
    int from = -1;
    for (int t = 0; t < T; ++t)
        if (idx[(int64_t)grp * T + t] == r) from = grp * T + t;
    for (int j = threadIdx.x; j < n; j += 128) dst[(int64_t)r * n + j] = from >= 0 ? src[(int64_t)from * n + j] : 0.f;
}

// ================================================================================================ grouped column sums
struct ColsumPack { dlsg_crit_colsum_desc d[DLSG_CRIT_COLSUM_MAX]; };
// grid (ceil(maxn / 64), count), 1024 threads = 16 row lanes x 64 columns; fixed order of additions
__global__ __launch_bounds__(1024) void crit_colsum_kernel(const ColsumPack pk) {
    __shared__ float red[16][64];
    const dlsg_crit_colsum_desc& d = pk.d[blockIdx.y];
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6, j = blockIdx.x * 64 + c;
    if (blockIdx.x * 64 >= d.n) return;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (j < d.n) {
        int r = rl;
        for (; r + 48 < d.rows; r += 64) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] += d.part[(int64_t)(r + 16 * u) * d.ld + j];
        }
        for (; r < d.rows; r += 16) acc[0] += d.part[(int64_t)r * d.ld + j];
        if (d.part_b)
            for (int r2 = rl; r2 < d.rows_b; r2 += 16) acc[1] += d.part_b[(int64_t)r2 * d.ld_b + j];
    }
    red[rl][c] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (rl == 0 && j < d.n) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][c];
        s *= d.scale;
        d.out[j] = s;
        if (d.out_b) d.out_b[j] = s;
    }
}

// `count` sums of slabs in one launch: out[i] = scale * sum_k src[k * stride + i], i < n (n % 4 == 0, 16-byte aligned): the K-split
// partial weight gradients of a critic update (critic.py _tn), folded in slab order.  A block owns 1024 consecutive elements of one
// descriptor; first[d] = the first block of descriptor d.
struct ReducePack { dlsg_crit_reduce_desc d[DLSG_CRIT_REDUCE_MAX]; int first[DLSG_CRIT_REDUCE_MAX + 1]; int count; };
__global__ __launch_bounds__(256) void crit_reduce_kernel(const ReducePack pk) {
    int di = 0;
    while (di + 1 < pk.count && (int)blockIdx.x >= pk.first[di + 1]) ++di;
    const dlsg_crit_reduce_desc& d = pk.d[di];
    const int64_t i = ((int64_t)(blockIdx.x - pk.first[di]) * 256 + threadIdx.x) * 4;
    if (i >= d.n) return;
    f32x4 acc = *reinterpret_cast<const f32x4*>(d.src + i);
    int k = 1;
    for (; k + 4 <= d.nslab; k += 4) {                 // four slabs per round trip; same order of additions
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(d.src + (int64_t)k * d.stride + i),
                    t1 = *reinterpret_cast<const f32x4*>(d.src + (int64_t)(k + 1) * d.stride + i),
                    t2 = *reinterpret_cast<const f32x4*>(d.src + (int64_t)(k + 2) * d.stride + i),
                    t3 = *reinterpret_cast<const f32x4*>(d.src + (int64_t)(k + 3) * d.stride + i);
        acc += t0; acc += t1; acc += t2; acc += t3;
    }
    for (; k < d.nslab; ++k) acc += *reinterpret_cast<const f32x4*>(d.src + (int64_t)k * d.stride + i);
    *reinterpret_cast<f32x4*>(d.out + i) = d.scale * acc;
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
bool al8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; }

}  // namespace

// ================================================================================================ entry points
extern "C" int dlsg_crit_embed_mix(const float* proj_tm, const int64_t* ids, const float* W, const float* bias, const float* eps, float* h,
                                   int ng, int B, int L, int V, void* stream) {
    if (!proj_tm || !bias || !h || (ng != 1 && ng != 3) || B < 1 || L < 1 || V < 1) return DLSG_EINVAL;
    if (ng == 3 && (!ids || !W || !eps)) return DLSG_EINVAL;
    if (!al16(proj_tm) || !al16(bias) || !al16(h)) return DLSG_EALIGN;
    hipLaunchKernelGGL(embed_mix_kernel, dim3(B * L), dim3(128), 0, ST(stream), proj_tm, ids, W, bias, eps, h, ng, B, L, V);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_crit_embed_mix_bwd(const float* ch, const float* eps, float* dhr, float* dhf_tm, int ng, int B, int L, void* stream) {
    if (!ch || !dhf_tm || (ng != 1 && ng != 3) || B < 1 || L < 1 || (ng == 3 && (!eps || !dhr))) return DLSG_EINVAL;
    if (!al16(ch) || !al16(dhf_tm) || !al16(dhr)) return DLSG_EALIGN;
    hipLaunchKernelGGL(embed_mix_bwd_kernel, dim3(B * L), dim3(128), 0, ST(stream), ch, eps, dhr, dhf_tm, ng, B, L);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_crit_vocab_scatter(const float* dhr, const int64_t* ids, float* dW, int rows, int V, void* stream) {
    if (!dhr || !ids || !dW || rows < 1 || V < 1 || rows > 12000) return DLSG_EINVAL;
    hipLaunchKernelGGL(vocab_scatter_kernel, dim3(rows, C / 64), dim3(256), (size_t)rows * sizeof(int), ST(stream), dhr, ids, dW, rows, V);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_crit_relu_taps(const float* x, const float* ref, const float* bias, float bias_scale, float* y, float* taps, int n,
                                   int L, void* stream) {
    if (!x || !ref || !y || !taps || n < 1 || L < 1) return DLSG_EINVAL;
    if (!al16(x) || !al16(ref) || !al16(y) || !al16(taps) || !al16(bias)) return DLSG_EALIGN;
    hipLaunchKernelGGL(relu_taps_kernel, dim3(n * L), dim3(128), 0, ST(stream), x, ref, bias, bias_scale, y, taps, L);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_crit_relu_taps_bwd(const float* dy, const float* dtaps, const float* ref, float* dx, int n, int L, void* stream) {
    if (!dy || !dtaps || !ref || !dx || n < 1 || L < 1) return DLSG_EINVAL;
    if (!al16(dy) || !al16(dtaps) || !al16(ref) || !al16(dx)) return DLSG_EALIGN;
    hipLaunchKernelGGL(relu_taps_bwd_kernel, dim3(n * L), dim3(128), 0, ST(stream), dy, dtaps, ref, dx, L);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}

extern "C" int64_t dlsg_cln_ws_floats(int rows, int N) { return (int64_t)2 * ln_blocks(rows) * N; }
extern "C" int dlsg_cln_fwd(const dlsg_cln_args* a, void* stream) {
    if (!cln_ok(a)) return DLSG_EINVAL;
    for (int g = 0; g < a->groups; ++g)
        if (!a->x[g] || !a->gamma[g] || !a->beta[g] || !a->y[g]) return DLSG_EINVAL;
    const dim3 grid(ln_blocks(a->rows), a->groups), block(256);
    CLN_DISPATCH(cln_fwd_kernel);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_cln_bwd(const dlsg_cln_args* a, void* stream) {
    if (!cln_ok(a) || a->ndy < 1 || a->ndy > 3) return DLSG_EINVAL;
    bool want = false;
    for (int g = 0; g < a->groups; ++g) {
        if (!a->x[g] || !a->gamma[g] || !a->dx[g]) return DLSG_EINVAL;
        for (int k = 0; k < a->ndy; ++k)
            if (!a->dy[k][g]) return DLSG_EINVAL;
        if ((a->dgamma[g] == nullptr) != (a->dbeta[g] == nullptr)) return DLSG_EINVAL;
        want = want || a->dgamma[g];
    }
    if ((want || a->defer) && !a->ws) return DLSG_EINVAL;
    const int nb = ln_blocks(a->rows);
    const dim3 grid(nb, a->groups), block(256);
    CLN_DISPATCH(cln_bwd_kernel);
    // defer: the per-workgroup partials stay in ws for the caller's own column sums (dlsg_crit_colsum)
    if (want && !a->defer) hipLaunchKernelGGL(cln_colsum_kernel, dim3(a->N / 64, 2, a->groups), block, 0, ST(stream), *a, nb, 2, 0);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_cln_bwd2(const dlsg_cln_args* a, void* stream) {
    if (!cln_ok(a) || a->ndy < 1 || a->ndy > 3 || !a->ws) return DLSG_EINVAL;
    for (int g = 0; g < a->groups; ++g) {
        if (!a->x[g] || !a->gamma[g] || !a->U[g] || !a->gx[g] || !a->gdy[g] || (!a->gpart[g] && !a->defer)) return DLSG_EINVAL;
        for (int k = 0; k < a->ndy; ++k)
            if (!a->dy[k][g]) return DLSG_EINVAL;
    }
    const int nb = ln_blocks(a->rows);
    const dim3 grid(nb, a->groups), block(256);
    CLN_DISPATCH(cln_bwd2_kernel);
    if (!a->defer) hipLaunchKernelGGL(cln_colsum_kernel, dim3(a->N / 64, 2, a->groups), block, 0, ST(stream), *a, nb, 2, 1);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}

static bool sa_ok(const dlsg_crit_sa_args* a) {
    return a && a->KQV && a->smask && a->n >= 1 && a->B >= 1 && a->L >= 1 && a->L <= LMAX && al8(a->KQV);
}
extern "C" int dlsg_crit_sa_fwd(const dlsg_crit_sa_args* a, void* stream) {
    if (!sa_ok(a) || !a->w || !a->ctx || !al8(a->ctx)) return DLSG_EINVAL;
    hipLaunchKernelGGL(sa_fwd_kernel, dim3(a->n), dim3(NT), 0, ST(stream), *a);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_crit_sa_bwd(const dlsg_crit_sa_args* a, void* stream) {
    if (!sa_ok(a) || !a->w || !a->dctx || !a->dKQV || !al8(a->dctx) || !al8(a->dKQV)) return DLSG_EINVAL;
    hipLaunchKernelGGL(sa_bwd_kernel, dim3(a->n), dim3(NT), 0, ST(stream), *a);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_crit_sa_bwd2(const dlsg_crit_sa_args* a, void* stream) {
    if (!sa_ok(a) || !a->w || !a->dctx || !a->U || !a->Uctx || !a->gKQV || !al8(a->dctx) || !al8(a->U) || !al8(a->Uctx) || !al8(a->gKQV))
        return DLSG_EINVAL;
    hipLaunchKernelGGL(sa_bwd2_kernel, dim3(a->n), dim3(NT), 0, ST(stream), *a);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}

static bool pattn_ok(const dlsg_crit_pattn_args* a) {
    if (!a || !a->smask || a->n < 1 || a->B < 1 || a->L < 1 || a->L > LMAX || a->T < 1 || a->T > TMAX) return false;
    for (int h = 0; h < 2; ++h)
        if (!a->a[h] || !a->e[h] || !al8(a->a[h]) || !al8(a->e[h])) return false;
    return true;
}
extern "C" int dlsg_crit_pattn_fwd(const dlsg_crit_pattn_args* a, void* stream) {
    if (!pattn_ok(a)) return DLSG_EINVAL;
    for (int h = 0; h < 2; ++h)
        if (!a->P[h] || !a->wgt[h] || !a->aggpre[h]) return DLSG_EINVAL;
    hipLaunchKernelGGL(pattn_fwd_kernel, dim3(a->n, 2), dim3(NT), 0, ST(stream), *a);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_crit_pattn_bwd(const dlsg_crit_pattn_args* a, void* stream) {
    if (!pattn_ok(a)) return DLSG_EINVAL;
    for (int h = 0; h < 2; ++h)
        if (!a->P[h] || !a->d_agg[h] || !a->d_wgt[h] || !a->da[h]) return DLSG_EINVAL;
    hipLaunchKernelGGL(pattn_bwd_kernel, dim3(a->n, 2), dim3(NT), 0, ST(stream), *a);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_crit_pattn_bwd2(const dlsg_crit_pattn_args* a, void* stream) {
    if (!pattn_ok(a)) return DLSG_EINVAL;
    for (int h = 0; h < 2; ++h)
        if (!a->P[h] || !a->d_agg[h] || !a->d_wgt[h] || !a->Ua[h] || !a->Uagg[h] || !a->Uwgt[h] || !a->ga[h] || !a->ge[h]) return DLSG_EINVAL;
    hipLaunchKernelGGL(pattn_bwd2_kernel, dim3(a->n, 2), dim3(NT), 0, ST(stream), *a);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}

static bool tsum_ok(const dlsg_crit_tsum_args* a) {
    return a && a->words && a->theta && a->gamma && a->beta && a->fusion && a->n >= 1 && a->L >= 1 && a->L <= LMAX && a->p >= 0.f &&
           a->p < 1.f && al8(a->words);
}
extern "C" int dlsg_crit_tsum_fwd(const dlsg_crit_tsum_args* a, void* stream) {
    if (!tsum_ok(a) || !a->adj || !a->u || !a->sent || !a->fus) return DLSG_EINVAL;
    hipLaunchKernelGGL(tsum_fwd_kernel, dim3(a->n), dim3(NT), 0, ST(stream), *a);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_crit_tsum_bwd(const dlsg_crit_tsum_args* a, void* stream) {
    if (!tsum_ok(a) || !a->d_fus || !a->dwords) return DLSG_EINVAL;
    hipLaunchKernelGGL(tsum_bwd_kernel, dim3(a->n), dim3(NT), 0, ST(stream), *a);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_crit_tsum_bwd2(const dlsg_crit_tsum_args* a, void* stream) {
    if (!tsum_ok(a) || !a->d_fus || !a->U || !a->Ufus || !a->gwords || !a->gpart) return DLSG_EINVAL;
    hipLaunchKernelGGL(tsum_bwd2_kernel, dim3(a->n), dim3(NT), 0, ST(stream), *a);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}

static bool score_ok(const dlsg_crit_score_args* a) {
    if (!a || !a->fus || a->n < 1 || a->B < 1 || a->T < 1 || a->T > TMAX || a->ng < 1 || a->ng > 8 || a->n != a->ng * a->B) return false;
    for (int h = 0; h < 2; ++h)
        if (!a->v[h] || !a->s[h] || !a->wc[h] || !a->wgt[h] || !a->pair[h] || !a->score[h]) return false;
    return true;
}
extern "C" int dlsg_crit_score_fwd(const dlsg_crit_score_args* a, void* stream) {
    if (!score_ok(a) || !a->both || !a->out || !a->bc[0] || !a->bc[1]) return DLSG_EINVAL;
    hipLaunchKernelGGL(score_pair_kernel, dim3(a->n, 2), dim3(NT), 0, ST(stream), *a);
    hipLaunchKernelGGL(score_out_kernel, dim3(1), dim3(NT), 0, ST(stream), *a);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_crit_score_bwd(const dlsg_crit_score_args* a, void* stream) {
    if (!score_ok(a) || !a->both || !a->d_out || !a->d_fus || !a->scratch) return DLSG_EINVAL;
    for (int h = 0; h < 2; ++h)
        if (!a->c_spre[h] || !a->d_wgt[h]) return DLSG_EINVAL;
    hipLaunchKernelGGL(score_bwd_head_kernel, dim3(1), dim3(NT), 0, ST(stream), *a);
    hipLaunchKernelGGL(score_bwd_kernel, dim3(a->n, 2), dim3(NT), 0, ST(stream), *a);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_crit_score_bwd2(const dlsg_crit_score_args* a, void* stream) {
    if (!score_ok(a) || a->ng != 1 || !a->d_out || !a->d_fus || !a->scratch || !a->Ufus || !a->dbc) return DLSG_EINVAL;
    for (int h = 0; h < 2; ++h)
        if (!a->c_spre[h] || !a->c_vpre[h] || !a->d_wgt[h] || !a->part_wc[h] || !a->Uspre[h] || !a->Uwgt[h]) return DLSG_EINVAL;
    hipLaunchKernelGGL(score_bwd2_a_kernel, dim3(a->n, 2), dim3(NT), 0, ST(stream), *a);
    hipLaunchKernelGGL(score_bwd2_b_kernel, dim3(1), dim3(NT), 0, ST(stream), *a);
    hipLaunchKernelGGL(score_bwd2_c_kernel, dim3(a->n, 2), dim3(NT), 0, ST(stream), *a);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}

extern "C" int dlsg_crit_gp(const float* g, const float* gG, const float* out, float* stats, float* vseed, float* gsc, float* q, int B,
                            int L, void* stream) {
    if (!g || !gG || !out || !stats || !vseed || !gsc || !q || B < 1 || L < 1) return DLSG_EINVAL;
    hipLaunchKernelGGL(gp_q_kernel, dim3(B), dim3(NT), 0, ST(stream), g, gG, q, L);
    hipLaunchKernelGGL(gp_out_kernel, dim3(B), dim3(NT), 0, ST(stream), g, gG, out, q, stats, vseed, gsc, B, L);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}

extern "C" int dlsg_crit_topk(const float* alpha, int64_t sa, int64_t lda, int na, const float* smask, int64_t* idx, int B, int L, int P,
                              int T, void* stream) {
    if (!alpha || !smask || !idx || B < 1 || L < 1 || P < 1 || P > 64 || T < 1 || T > P || na < P) return DLSG_EINVAL;
    hipLaunchKernelGGL(topk_kernel, dim3(B, 2), dim3(64), 0, ST(stream), alpha, sa, lda, na, smask, idx, B, L, P, T);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_crit_unselect(const float* src, const int64_t* idx, float* dst, int rows_src, int rows_dst, int per, int n, void* stream) {
    if (!src || !idx || !dst || rows_src < 1 || rows_dst < 1 || per < 1 || rows_dst % per || n < 1) return DLSG_EINVAL;
    const int groups = rows_dst / per;
    if (rows_src % groups) return DLSG_EINVAL;
    hipLaunchKernelGGL(unselect_kernel, dim3(rows_dst), dim3(128), 0, ST(stream), src, idx, dst, rows_src / groups, per, n);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}

extern "C" int dlsg_crit_reduce(const dlsg_crit_reduce_desc* d, int count, void* stream) {
    if (!d || count < 1 || count > DLSG_CRIT_REDUCE_MAX) return DLSG_EINVAL;
    ReducePack pk;
    int blocks = 0;
    for (int i = 0; i < count; ++i) {
        if (!d[i].src || !d[i].out || d[i].nslab < 1 || d[i].n < 4 || (d[i].n & 3) || (d[i].stride & 3) || !al16(d[i].src) || !al16(d[i].out))
            return DLSG_EINVAL;
        pk.d[i] = d[i];
        pk.first[i] = blocks;
        blocks += (int)((d[i].n / 4 + 255) / 256);
    }
    pk.first[count] = blocks;
    pk.count = count;
    hipLaunchKernelGGL(crit_reduce_kernel, dim3(blocks), dim3(256), 0, ST(stream), pk);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}

extern "C" int dlsg_crit_colsum(const dlsg_crit_colsum_desc* d, int count, void* stream) {
    if (!d || count < 1 || count > DLSG_CRIT_COLSUM_MAX) return DLSG_EINVAL;
    ColsumPack pk;
    int maxn = 0;
    for (int i = 0; i < count; ++i) {
        if (!d[i].part || !d[i].out || d[i].rows < 1 || d[i].n < 1 || d[i].n > 2048 || (d[i].part_b && d[i].rows_b < 1)) return DLSG_EINVAL;
        pk.d[i] = d[i];
        if (d[i].n > maxn) maxn = d[i].n;
    }
    hipLaunchKernelGGL(crit_colsum_kernel, dim3((maxn + 63) / 64, count), dim3(1024), 0, ST(stream), pk);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
