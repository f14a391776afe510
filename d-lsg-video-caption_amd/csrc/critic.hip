// Pointwise kernels of the DiscV2 critic (models/model.py:110-168, run_gun.py:339-398).
//
// The critic's LSTM (nn.LSTM(512, 512), model.py:122) is differentiated TWICE per critic update: the WGAN-GP gradient penalty
// takes d(score)/d(input) with create_graph=True and the loss backward then runs through that gradient (run_gun.py:362-371).
// Unrolled over ATen ops a cell step is ~11 launches forward, ~20 in its backward and ~40 in the backward of the backward,
// x 26 steps: ~2 600 of the ~3 000 launches of one critic update, each at the ~4 us launch floor.  Here the cell's pointwise
// part is ONE launch per level; the recurrent products stay ordinary matmuls, which autograd differentiates itself.
//
//   level 0  (h, c)          = cell(a, c_prev)                         a = x W_ih^T + h_prev W_hh^T + b, gates i, f, g, o
//   level 1  (da, dc_prev)   = cell'(a, c_prev; dh, dc)                the cell's backward
//   level 2  (ga, gc_prev, gdh, gdc) = vector-Jacobian product of level 1 w.r.t. all four of its inputs, cotangents (u, uc)
//
// With i = s(a_i), f = s(a_f), g = tanh(a_g), o = s(a_o), c = f c_prev + i g, tc = tanh(c), q = 1 - tc^2,
// s_i = i(1-i), s_f = f(1-f), s_o = o(1-o), s_g = 1 - g^2:
//   level 1:  dct = dc + dh o q;  da_i = dct g s_i;  da_f = dct c_prev s_f;  da_g = dct i s_g;  da_o = dh tc s_o;  dc_prev = dct f
//   level 2:  A  = u_i g s_i + u_f c_prev s_f + u_g i s_g + uc f            (= dL/d dct)
//             Gc = q (u_o dh s_o - 2 A dh o tc)                              (= dL/dc through tc)
//             ga_i = dct s_i (u_i g (1-2i) + u_g s_g) + Gc g s_i
//             ga_f = dct s_f (u_f c_prev (1-2f) + uc) + Gc c_prev s_f
//             ga_g = dct s_g (u_i s_i - 2 u_g i g) + Gc i s_g
//             ga_o = s_o dh (A q + u_o tc (1-2o))
//             gc_prev = dct u_f s_f + Gc f;   gdh = A o q + u_o tc s_o;   gdc = A
// (checked against autograd's own double backward of the unrolled cell in tests/test_gpu_ops.py and tests/test_engine_host_logic.py)
#include "common.hpp"
#include "dlsg.h"

namespace {

struct Gates {
    float i, f, g, o, c, tc, q;
};
__device__ __forceinline__ Gates gates_of(const float* __restrict__ a, int64_t lda, int r, int j, int H, float cp) {
    const float* ar = a + (int64_t)r * lda + j;
    Gates x;
    x.i = dlsg::sigmoidf_(ar[0]);
    x.f = dlsg::sigmoidf_(ar[H]);
    x.g = tanhf(ar[2 * H]);
    x.o = dlsg::sigmoidf_(ar[3 * H]);
    x.c = x.f * cp + x.i * x.g;
    x.tc = tanhf(x.c);
    x.q = 1.f - x.tc * x.tc;
    return x;
}

__global__ __launch_bounds__(256) void cell_fwd_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ c_prev,
                                                       float* __restrict__ h, float* __restrict__ c, int rows, int H) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * H) return;
    const int r = idx / H, j = idx - r * H;
    const Gates x = gates_of(a, lda, r, j, H, c_prev ? c_prev[idx] : 0.f);
    c[idx] = x.c;
    h[idx] = x.o * x.tc;
}

__global__ __launch_bounds__(256) void cell_bwd_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ c_prev,
                                                       const float* __restrict__ dh, const float* __restrict__ dc,
                                                       float* __restrict__ da, float* __restrict__ dc_prev, int rows, int H) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * H) return;
    const int r = idx / H, j = idx - r * H;
    const float cp = c_prev[idx];
    const Gates x = gates_of(a, lda, r, j, H, cp);
    const float dhv = dh[idx];
    const float dct = dc[idx] + dhv * x.o * x.q;
    float* d = da + (int64_t)r * 4 * H + j;
    d[0] = dct * x.g * x.i * (1.f - x.i);
    d[H] = dct * cp * x.f * (1.f - x.f);
    d[2 * H] = dct * x.i * (1.f - x.g * x.g);
    d[3 * H] = dhv * x.tc * x.o * (1.f - x.o);
    dc_prev[idx] = dct * x.f;
}

// the backward step of the whole-sequence op: dh = dh1 (+ dh2), dc = (dc1) (+ dc2), da = cell'(...) (+ da_inj); the summed
// dh / dc are kept (dh_tot, dc_tot): they are the cell inputs the backward of this step is differentiated at
__global__ __launch_bounds__(256) void cell_bwd_seq_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ c_prev,
                                                           const float* __restrict__ dh1, const float* __restrict__ dh2,
                                                           const float* __restrict__ dc1, const float* __restrict__ dc2,
                                                           const float* __restrict__ da_inj, float* __restrict__ da,
                                                           float* __restrict__ dc_prev, float* __restrict__ dh_tot,
                                                           float* __restrict__ dc_tot, int rows, int H) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * H) return;
    const int r = idx / H, j = idx - r * H;
    const float cp = c_prev ? c_prev[idx] : 0.f;
    const Gates x = gates_of(a, lda, r, j, H, cp);
    const float dhv = dh1[idx] + (dh2 ? dh2[idx] : 0.f);
    const float dcv = (dc1 ? dc1[idx] : 0.f) + (dc2 ? dc2[idx] : 0.f);
    dh_tot[idx] = dhv;
    dc_tot[idx] = dcv;
    const float dct = dcv + dhv * x.o * x.q;
    float* d = da + (int64_t)r * 4 * H + j;
    float v0 = dct * x.g * x.i * (1.f - x.i), v1 = dct * cp * x.f * (1.f - x.f), v2 = dct * x.i * (1.f - x.g * x.g),
          v3 = dhv * x.tc * x.o * (1.f - x.o);
    if (da_inj) {
        const float* q = da_inj + (int64_t)r * 4 * H + j;
        v0 += q[0]; v1 += q[H]; v2 += q[2 * H]; v3 += q[3 * H];
    }
    d[0] = v0; d[H] = v1; d[2 * H] = v2; d[3 * H] = v3;
    dc_prev[idx] = dct * x.f;
}

__global__ __launch_bounds__(256) void cell_bwd2_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ c_prev,
                                                        const float* __restrict__ dh, const float* __restrict__ dc,
                                                        const float* __restrict__ u, const float* __restrict__ uc,
                                                        float* __restrict__ ga, float* __restrict__ gc_prev,
                                                        float* __restrict__ gdh, float* __restrict__ gdc, int rows, int H) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * H) return;
    const int r = idx / H, j = idx - r * H;
    const float cp = c_prev ? c_prev[idx] : 0.f;
    const Gates x = gates_of(a, lda, r, j, H, cp);
    const float dhv = dh[idx];
    const float* ur = u + (int64_t)r * 4 * H + j;
    const float ui = ur[0], uf = ur[H], ug = ur[2 * H], uo = ur[3 * H], ucv = uc ? uc[idx] : 0.f;
    const float si = x.i * (1.f - x.i), sf = x.f * (1.f - x.f), so = x.o * (1.f - x.o), sg = 1.f - x.g * x.g;
    const float dct = dc[idx] + dhv * x.o * x.q;
    const float A = ui * x.g * si + uf * cp * sf + ug * x.i * sg + ucv * x.f;
    const float Gc = x.q * (uo * dhv * so - 2.f * A * dhv * x.o * x.tc);
    float* g = ga + (int64_t)r * 4 * H + j;
    g[0] = dct * si * (ui * x.g * (1.f - 2.f * x.i) + ug * sg) + Gc * x.g * si;
    g[H] = dct * sf * (uf * cp * (1.f - 2.f * x.f) + ucv) + Gc * cp * sf;
    g[2 * H] = dct * sg * (ui * si - 2.f * ug * x.i * x.g) + Gc * x.i * sg;
    g[3 * H] = so * dhv * (A * x.q + uo * x.tc * (1.f - 2.f * x.o));
    if (gc_prev) gc_prev[idx] = dct * uf * sf + Gc * x.f;
    gdh[idx] = A * x.o * x.q + uo * x.tc * so;
    gdc[idx] = A;
}

// ------------------------------------------------------------------------------------------------ (tanh +) LayerNorm, three levels
// y = LN(t) gamma + beta over rows of N, t = tanh(x) or x -- the nine normalisations of the critic (models/model.py:124-131,
// layer.py:661-689).  On ATen ops one instance costs ~85 launches across forward, backward and the backward of the backward
// (the double backward of native_layer_norm_backward is a ~40-op composite): ~770 of a critic update's ~2 000.  Here:
//   level 0  y
//   level 1  (dx, dgamma, dbeta) from dy:        a = dy gamma, m1 = mean(a), m2 = mean(a n), dt = r (a - m1 - n m2), dx = dt s
//   level 2  (gx, ggamma, gdy) from cotangents (U on dx, vg on dgamma, vb on dbeta):
//            W = U s, w1 = mean(W), w2 = mean(W n), core = r (W - w1 - n w2)
//            gdy = gamma core + vg n + vb;   ggamma = sum_rows dy core
//            Q = sum(W dt) / r,  Pn = -r (W m2 + a w2) + vg dy,  p1 = mean(Pn), p2 = mean(Pn n)
//            gx = s [ r (Pn - p1 - n p2) - Q r^2 n / N  - 2 t U dt ]        (last term only with the tanh)
// with n = (t - mean t) r, r = rsqrt(var t + eps), s = 1 - t^2 (1 without the tanh).  One wave per row (N = 64 E, E <= 16,
// lane l holds columns l + 64 e: coalesced), statistics recomputed from x at every level; column sums (dgamma, dbeta,
// ggamma) are per-workgroup partials in a caller workspace, summed in a fixed order by a second launch.
constexpr int LN_MAXE = 16;

struct RowStats {
    float mu, r;
};
template <bool TANH, int EMAX>
__device__ __forceinline__ RowStats load_row(const float* __restrict__ xr, int E, int lane, float eps, float (&t)[EMAX]) {
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < EMAX; ++e)
        if (e < E) {
            const float v = xr[lane + 64 * e];
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

template <bool TANH, int EMAX>
__global__ __launch_bounds__(256) void tanh_ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ y, int rows, int E,
                                                          float eps) {
    const int lane = threadIdx.x & 63, gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4, N = 64 * E;
    // blockIdx.y = row group: `rows` consecutive rows with their own gamma / beta (several same-shape LayerNorms in one launch)
    x += (int64_t)blockIdx.y * rows * N; y += (int64_t)blockIdx.y * rows * N;
    gamma += blockIdx.y * N; beta += blockIdx.y * N;
    float t[EMAX];
    for (int row = gw; row < rows; row += nw) {
        const RowStats st = load_row<TANH, EMAX>(x + (int64_t)row * N, E, lane, eps, t);
#pragma unroll
        for (int e = 0; e < EMAX; ++e)
            if (e < E) {
                const int j = lane + 64 * e;
                y[(int64_t)row * N + j] = (t[e] - st.mu) * st.r * gamma[j] + beta[j];
            }
    }
}

template <bool TANH, int EMAX>
__global__ __launch_bounds__(256) void tanh_ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ dy, float* __restrict__ dx,
                                                          float* __restrict__ part, int rows, int E, float eps) {
    const int lane = threadIdx.x & 63, gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4, N = 64 * E;
    x += (int64_t)blockIdx.y * rows * N; dy += (int64_t)blockIdx.y * rows * N; dx += (int64_t)blockIdx.y * rows * N;
    gamma += blockIdx.y * N;
    part += (int64_t)blockIdx.y * 2 * gridDim.x * N;                    // [group][dgamma | dbeta][workgroup][N]
    float t[EMAX], gam[EMAX], pg[EMAX], pb[EMAX];
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
        pg[e] = pb[e] = 0.f;
        gam[e] = e < E ? gamma[lane + 64 * e] : 0.f;
    }
    for (int row = gw; row < rows; row += nw) {
        const RowStats st = load_row<TANH, EMAX>(x + (int64_t)row * N, E, lane, eps, t);
        float a[EMAX], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < EMAX; ++e)
            if (e < E) {
                const float d = dy[(int64_t)row * N + lane + 64 * e];
                const float n = (t[e] - st.mu) * st.r;
                a[e] = d * gam[e];
                s1 += a[e];
                s2 += a[e] * n;
                pg[e] += d * n;
                pb[e] += d;
            }
        const float inv = 1.f / (64.f * E);
        const float m1 = dlsg::wave_sum(s1) * inv, m2 = dlsg::wave_sum(s2) * inv;
#pragma unroll
        for (int e = 0; e < EMAX; ++e)
            if (e < E) {
                const float n = (t[e] - st.mu) * st.r;
                const float dt = st.r * (a[e] - m1 - n * m2);
                dx[(int64_t)row * N + lane + 64 * e] = TANH ? dt * (1.f - t[e] * t[e]) : dt;
            }
    }
    // per-workgroup column partials: the four waves' sums meet in LDS, one (N)-row per quantity and workgroup leaves
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
__global__ __launch_bounds__(256) void tanh_ln_bwd2_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ dy, const float* __restrict__ U,
                                                           const float* __restrict__ vg, const float* __restrict__ vb,
                                                           float* __restrict__ gx, float* __restrict__ gdy,
                                                           float* __restrict__ part, int rows, int E, float eps) {
    const int lane = threadIdx.x & 63, gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4, N = 64 * E;
    {
        const int64_t o = (int64_t)blockIdx.y * rows * N;
        x += o; dy += o; U += o; gx += o; gdy += o;
        gamma += blockIdx.y * N; vg += blockIdx.y * N; vb += blockIdx.y * N;
        part += (int64_t)blockIdx.y * gridDim.x * N;                    // [group][workgroup][N]
    }
    float t[EMAX], gam[EMAX], vgl[EMAX], vbl[EMAX], pgg[EMAX];
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
        pgg[e] = 0.f;
        gam[e] = e < E ? gamma[lane + 64 * e] : 0.f;
        vgl[e] = e < E ? vg[lane + 64 * e] : 0.f;
        vbl[e] = e < E ? vb[lane + 64 * e] : 0.f;
    }
    const float inv = 1.f / (64.f * E);
    for (int row = gw; row < rows; row += nw) {
        const RowStats st = load_row<TANH, EMAX>(x + (int64_t)row * N, E, lane, eps, t);
        float d[EMAX], W[EMAX], Uv[EMAX], s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
#pragma unroll
        for (int e = 0; e < EMAX; ++e)
            if (e < E) {
                const int64_t o = (int64_t)row * N + lane + 64 * e;
                d[e] = dy[o];
                const float u = Uv[e] = U[o];
                const float n = (t[e] - st.mu) * st.r, a = d[e] * gam[e];
                W[e] = TANH ? u * (1.f - t[e] * t[e]) : u;
                s1 += a;
                s2 += a * n;
                s3 += W[e];
                s4 += W[e] * n;
            }
        const float m1 = dlsg::wave_sum(s1) * inv, m2 = dlsg::wave_sum(s2) * inv;
        const float w1 = dlsg::wave_sum(s3) * inv, w2 = dlsg::wave_sum(s4) * inv;
        float Pn[EMAX], q = 0.f, s5 = 0.f, s6 = 0.f;
#pragma unroll
        for (int e = 0; e < EMAX; ++e)
            if (e < E) {
                const float n = (t[e] - st.mu) * st.r, a = d[e] * gam[e];
                const float dt = st.r * (a - m1 - n * m2);
                q += W[e] * dt;
                Pn[e] = -st.r * (W[e] * m2 + a * w2) + vgl[e] * d[e];
                s5 += Pn[e];
                s6 += Pn[e] * n;
                const float core = st.r * (W[e] - w1 - n * w2);
                gdy[(int64_t)row * N + lane + 64 * e] = gam[e] * core + vgl[e] * n + vbl[e];
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
                    const float a = d[e] * gam[e];
                    const float dt = st.r * (a - m1 - n * m2);
                    G = (G - 2.f * t[e] * Uv[e] * dt) * s;
                }
                gx[(int64_t)row * N + lane + 64 * e] = G;
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

// out_k[g][j] = sum_w part[g][k][w][j], w in order (deterministic); grid (N / 64, K, groups)
__global__ __launch_bounds__(256) void ln_colsum_kernel(const float* __restrict__ part, int nw, int N, float* __restrict__ o0,
                                                        float* __restrict__ o1) {
    __shared__ float red[4][64];
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6, j = blockIdx.x * 64 + c;
    const float* p = part + ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * nw * N;
    o0 += blockIdx.z * N; o1 += blockIdx.z * N;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int w = q;
    for (; w + 28 < nw; w += 32) {                       // 8 independent loads in flight per thread
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] += p[(int64_t)(w + 4 * u) * N + j];
    }
    for (; w < nw; w += 4) acc[0] += p[(int64_t)w * N + j];
    red[q][c] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (q == 0) (blockIdx.y == 0 ? o0 : o1)[j] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// ------------------------------------------------------------------------------------------------ Conv1d(k = 3) taps
// DiscV2's ResBlock convolution (sublayer.py:107-119, Conv1d(512, 512, 3, padding = 1) over the word axis) is ONE product on the
// three shifted copies of the sequence:  taps[b, t, k C + c] = x[b, t + k - 1, c]  (zero outside the caption).  Built from ATen
// ops (pad + three slices + cat) that is 5 launches forward and ~10 per backward level (every slice gradient is a zero-filled
// (n, L + 2, C) array plus a copy); here it is one launch, and so is its adjoint
//    dx[b, t, c] = sum_k d[b, t - k + 1, k C + c],
// the two being each other's backward (both are linear maps).
__global__ __launch_bounds__(256) void conv_taps_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int L, int C4,
                                                        int adjoint) {
    const int64_t total = (int64_t)n * L * (adjoint ? C4 : 3 * C4);
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    f32x4* y4 = reinterpret_cast<f32x4*>(y);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        if (!adjoint) {
            const int c = (int)(i % C4), k = (int)((i / C4) % 3);
            const int64_t row = i / (3 * C4);
            const int t = (int)(row % L) + k - 1;
            y4[i] = (t >= 0 && t < L) ? x4[(row + k - 1) * C4 + c] : f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
            const int c = (int)(i % C4);
            const int64_t row = i / C4;
            const int t = (int)(row % L);
            f32x4 acc = x4[row * 3 * C4 + C4 + c];                                   // k = 1: d[b, t]
            if (t + 1 < L) acc += x4[(row + 1) * 3 * C4 + c];                        // k = 0: d[b, t + 1]
            if (t > 0) acc += x4[(row - 1) * 3 * C4 + 2 * C4 + c];                   // k = 2: d[b, t - 1]
            y4[i] = acc;
        }
    }
}

inline int ln_blocks(int rows) { return rows < 2048 ? (rows + 3) / 4 : 512; }

// rows of up to 512 columns keep 8 values per lane in registers, wider ones 16
#define LN_DISPATCH(KERNEL, ...)                                                                                    \
    do {                                                                                                            \
        if (N <= 512) {                                                                                             \
            if (pre_tanh) hipLaunchKernelGGL((KERNEL<true, 8>), grid, block, 0, ST(stream), __VA_ARGS__);           \
            else hipLaunchKernelGGL((KERNEL<false, 8>), grid, block, 0, ST(stream), __VA_ARGS__);                   \
        } else {                                                                                                    \
            if (pre_tanh) hipLaunchKernelGGL((KERNEL<true, 16>), grid, block, 0, ST(stream), __VA_ARGS__);          \
            else hipLaunchKernelGGL((KERNEL<false, 16>), grid, block, 0, ST(stream), __VA_ARGS__);                  \
        }                                                                                                           \
    } while (0)

inline hipStream_t ST(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace

extern "C" int dlsg_lstm_cell_fwd(const float* a, int64_t lda, const float* c_prev, float* h, float* c, int rows, int H, void* stream) {
    if (!a || !h || !c || rows < 0 || H < 1 || lda < 4 * (int64_t)H) return DLSG_EINVAL;     /* c_prev NULL: zero state */
    if (rows == 0) return DLSG_OK;
    hipLaunchKernelGGL(cell_fwd_kernel, dim3((rows * H + 255) / 256), dim3(256), 0, ST(stream), a, lda, c_prev, h, c, rows, H);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_lstm_cell_bwd(const float* a, int64_t lda, const float* c_prev, const float* dh, const float* dc, float* da,
                                  float* dc_prev, int rows, int H, void* stream) {
    if (!a || !c_prev || !dh || !dc || !da || !dc_prev || rows < 0 || H < 1 || lda < 4 * (int64_t)H) return DLSG_EINVAL;
    if (rows == 0) return DLSG_OK;
    hipLaunchKernelGGL(cell_bwd_kernel, dim3((rows * H + 255) / 256), dim3(256), 0, ST(stream), a, lda, c_prev, dh, dc, da, dc_prev,
                       rows, H);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_lstm_cell_bwd_seq(const float* a, int64_t lda, const float* c_prev, const float* dh1, const float* dh2, const float* dc1,
                                      const float* dc2, const float* da_inj, float* da, float* dc_prev, float* dh_tot, float* dc_tot,
                                      int rows, int H, void* stream) {
    if (!a || !dh1 || !da || !dc_prev || !dh_tot || !dc_tot || rows < 0 || H < 1 || lda < 4 * (int64_t)H) return DLSG_EINVAL;
    if (rows == 0) return DLSG_OK;
    hipLaunchKernelGGL(cell_bwd_seq_kernel, dim3((rows * H + 255) / 256), dim3(256), 0, ST(stream), a, lda, c_prev, dh1, dh2, dc1, dc2,
                       da_inj, da, dc_prev, dh_tot, dc_tot, rows, H);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_lstm_cell_bwd2(const float* a, int64_t lda, const float* c_prev, const float* dh, const float* dc, const float* u,
                                   const float* uc, float* ga, float* gc_prev, float* gdh, float* gdc, int rows, int H, void* stream) {
    if (!a || !dh || !dc || !u || !ga || !gdh || !gdc || rows < 0 || H < 1 || lda < 4 * (int64_t)H)
        return DLSG_EINVAL;                                  /* c_prev, uc NULL: zeros; gc_prev NULL: not wanted */
    if (rows == 0) return DLSG_OK;
    hipLaunchKernelGGL(cell_bwd2_kernel, dim3((rows * H + 255) / 256), dim3(256), 0, ST(stream), a, lda, c_prev, dh, dc, u, uc, ga,
                       gc_prev, gdh, gdc, rows, H);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}

extern "C" int dlsg_conv_taps(const float* x, float* y, int n, int L, int C, int adjoint, void* stream) {
    if (!x || !y || n < 0 || L < 1 || C < 4 || (C & 3) || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15)) return DLSG_EINVAL;
    if (n == 0) return DLSG_OK;
    const int64_t total = (int64_t)n * L * (adjoint ? C / 4 : 3 * (C / 4));
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(conv_taps_kernel, dim3(blocks), dim3(256), 0, ST(stream), x, y, n, L, C / 4, adjoint);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}

extern "C" int64_t dlsg_tanh_ln_ws_floats(int rows, int N) { return (int64_t)2 * ln_blocks(rows) * N; }

// groups > 1: `groups` consecutive blocks of `rows` rows, block g normalised with gamma[g], beta[g] ((groups, N) arrays) -- the
// same-shape LayerNorms of the critic's two proposal scorers in one launch.  ws: groups * dlsg_tanh_ln_ws_floats(rows, N).
extern "C" int dlsg_tanh_ln_fwd(const float* x, const float* gamma, const float* beta, float* y, int rows, int N, float eps,
                                int pre_tanh, int groups, void* stream) {
    if (!x || !gamma || !beta || !y || rows < 0 || N < 64 || N % 64 || N > 64 * LN_MAXE || groups < 1 || groups > 64) return DLSG_EINVAL;
    if (rows == 0) return DLSG_OK;
    const dim3 grid(ln_blocks(rows), groups), block(256);
    LN_DISPATCH(tanh_ln_fwd_kernel, x, gamma, beta, y, rows, N / 64, eps);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_tanh_ln_bwd(const float* x, const float* gamma, const float* dy, float* dx, float* dgamma, float* dbeta,
                                float* ws, int rows, int N, float eps, int pre_tanh, int groups, void* stream) {
    if (!x || !gamma || !dy || !dx || !dgamma || !dbeta || !ws || rows < 1 || N < 64 || N % 64 || N > 64 * LN_MAXE || groups < 1 ||
        groups > 64)
        return DLSG_EINVAL;
    const int nb = ln_blocks(rows);
    const dim3 grid(nb, groups), block(256);
    LN_DISPATCH(tanh_ln_bwd_kernel, x, gamma, dy, dx, ws, rows, N / 64, eps);
    hipLaunchKernelGGL(ln_colsum_kernel, dim3(N / 64, 2, groups), block, 0, ST(stream), ws, nb, N, dgamma, dbeta);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_tanh_ln_bwd2(const float* x, const float* gamma, const float* dy, const float* U, const float* vg, const float* vb,
                                 float* gx, float* ggamma, float* gdy, float* ws, int rows, int N, float eps, int pre_tanh,
                                 int groups, void* stream) {
    if (!x || !gamma || !dy || !U || !vg || !vb || !gx || !ggamma || !gdy || !ws || rows < 1 || N < 64 || N % 64 || N > 64 * LN_MAXE ||
        groups < 1 || groups > 64)
        return DLSG_EINVAL;
    const int nb = ln_blocks(rows);
    const dim3 grid(nb, groups), block(256);
    LN_DISPATCH(tanh_ln_bwd2_kernel, x, gamma, dy, U, vg, vb, gx, gdy, ws, rows, N / 64, eps);
    hipLaunchKernelGGL(ln_colsum_kernel, dim3(N / 64, 1, groups), block, 0, ST(stream), ws, nb, N, ggamma, ggamma);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
