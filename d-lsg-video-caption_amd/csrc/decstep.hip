// Fused decoder-step kernels (Decoder.decode, reference models/layer.py:569-602).
//
// A word step of the decoder is two gate GEMMs with, between and after them, a chain of per-batch-row operations that the
// unfused schedule runs as 6 forward launches of 8-10 us each (mostly launch + memory round-trip latency on 64 rows).
// Everything in that chain is independent across batch rows, so one workgroup per row runs the whole chain with the
// intermediate vectors in LDS:
//   dec_mid : query LSTM cell pointwise -> LayerNorm(+dropout) -> attention over cached K', V' (both streams)
//             -> tanh -> LayerNorm(+dropout)
//   dec_tail: language LSTM cell pointwise (+dropout) -> tanh(LayerNorm) for the vocab projection
// Arithmetic is the same as lstm_pw_fwd / rowln_fwd / decatt_fwd (rowops.hip, attention.hip), element for element.
#include "common.hpp"
#include "dlsg.h"

using namespace dlsg;

namespace {

constexpr int DT = 1024;         // threads per row: one workgroup has to hide the whole chain's memory latency itself
constexpr int MAXW = 2048;       // max Q / H / D
constexpr int MAXP = 72;        // attended rows per stream: proposals, or the frame nodes of the baseline decoders (PositionalEncoding max_len, sublayer.py:87)

template <int V> struct Vec;
template <> struct Vec<4> { using T = float4; };
template <> struct Vec<1> { using T = float; };
template <int V> __device__ __forceinline__ void vload(float (&r)[V], const float* p) {
    if constexpr (V == 4) { const float4 t = *reinterpret_cast<const float4*>(p); r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w; }
    else r[0] = *p;
}
template <int V> __device__ __forceinline__ void vstore(float* p, const float (&r)[V]) {
    if constexpr (V == 4) *reinterpret_cast<float4*>(p) = make_float4(r[0], r[1], r[2], r[3]);
    else *p = r[0];
}

// two block-wide sums at once (same contract as block_sum; `red` >= 32 floats)
__device__ __forceinline__ void block_sum2(float& a, float& b, float* red) {
    a = wave_sum(a); b = wave_sum(b);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) { red[w] = a; red[16 + w] = b; }
    __syncthreads();
    float ra = 0.f, rb = 0.f;
#pragma unroll
    for (int i = 0; i < DT / 64; ++i) { ra += red[i]; rb += red[16 + i]; }
    a = ra; b = rb;
}

// LSTM cell of one row (gate order i, f, g, o): gate pre-activations = slabs + addend + biases, all loads of a thread
// independent (V-wide, slabs unrolled by 4) so one round trip covers them; activated gates go to `gb` (LDS, 4N) and
// `gates`; then c = f*c_prev + i*g, h = o*tanh(c) (* dropout) per unit; h is left in `hb` (LDS).
template <int V>
__device__ __forceinline__ void cell_row(int b, int N, const float* slabs, int nslab, int64_t slab_stride, const float* addend,
                                         int64_t ldadd, const float* b_ih, const float* b_hh, const float* c_prev, float* c_out,
                                         float* h_out, float* gates, float p, uint32_t site, uint64_t seed, float* gb, float* hb) {
    // (c_prev is needed after the gates: requested with them, not a round trip later)
    float cpv[MAXW / DT];
#pragma unroll
    for (int i = 0; i < MAXW / DT; ++i) {
        const int j = threadIdx.x + i * DT;
        cpv[i] = (c_prev && j < N) ? c_prev[(int64_t)b * N + j] : 0.f;
    }
    for (int col = threadIdx.x * V; col < 4 * N; col += DT * V) {
        float acc[V], t[4][V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        const float* sp = slabs + (int64_t)b * 4 * N + col;
        int k = 0;
        for (; k + 4 <= nslab; k += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) vload<V>(t[u], sp + (k + u) * slab_stride);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < V; ++e) acc[e] += t[u][e];
        }
        if (k < nslab) {                                   // the last 1-3 slabs: requested together as well (the query gates come in 3)
            const int rem = nslab - k;
#pragma unroll
            for (int u = 0; u < 3; ++u)
                if (u < rem) vload<V>(t[u], sp + (k + u) * slab_stride);
#pragma unroll
            for (int u = 0; u < 3; ++u)
                if (u < rem)
#pragma unroll
                    for (int e = 0; e < V; ++e) acc[e] += t[u][e];
        }
        if (addend) { vload<V>(t[0], addend + (int64_t)b * ldadd + col);
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] += t[0][e]; }
        if (b_ih) { vload<V>(t[1], b_ih + col);
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] += t[1][e]; }
        if (b_hh) { vload<V>(t[2], b_hh + col);
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] += t[2][e]; }
        const bool is_g = (col / N) == 2;
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = is_g ? tanhf(acc[e]) : sigmoidf_(acc[e]);
        vstore<V>(gates + (int64_t)b * 4 * N + col, acc);
        vstore<V>(gb + col, acc);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MAXW / DT; ++i) {
        const int j = threadIdx.x + i * DT;
        if (j >= N) break;
        const float cp = cpv[i];
        const float c = gb[N + j] * cp + gb[j] * gb[2 * N + j];
        float h = gb[3 * N + j] * tanhf(c);
        c_out[(int64_t)b * N + j] = c;
        if (p > 0.f) h *= drop_scale(seed, site, (uint64_t)b * N + j, p);
        h_out[(int64_t)b * N + j] = h;
        hb[j] = h;
    }
    __syncthreads();
}

// LayerNorm statistics of a vector held in LDS (two-pass, like rowln_fwd_kernel)
__device__ __forceinline__ void ln_stats(const float* v, int n, float eps, float* red, float& mean, float& rstd) {
    float s = 0.f;
    for (int j = threadIdx.x; j < n; j += DT) s += v[j];
    mean = block_sum(s, red) / n;
    float q = 0.f;
    for (int j = threadIdx.x; j < n; j += DT) { const float d = v[j] - mean; q += d * d; }
    rstd = rsqrtf(block_sum(q, red) / n + eps);
}

template <int V>
__global__ __launch_bounds__(DT) void dec_mid_fwd_kernel(const dlsg_dec_mid_args a) {
    __shared__ __attribute__((aligned(16))) float gb[4 * MAXW];   // activated gates; later tanh(context) of both streams
    __shared__ __attribute__((aligned(16))) float hb[MAXW];       // h
    __shared__ __attribute__((aligned(16))) float qb[MAXW];       // q_cur = dropout(LN(h)): the attention query
    __shared__ float sc[2 * MAXP], wt[2 * MAXP];
    __shared__ float red[32];
    const int b = blockIdx.x;
    const int bk = a.kv_div > 1 ? b / a.kv_div : b;          // the block of K', V' this row attends over
    const int Q = a.Q, H = a.H, P = a.P, ns = a.nstream;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t seed = a.seed + (a.seed_ptr ? *a.seed_ptr : 0ull);

    cell_row<V>(b, Q, a.slabs, a.nslab, a.slab_stride, a.addend, a.ldadd, a.b_ih, a.b_hh, a.c_prev, a.c, a.h, a.gates, 0.f, 0,
                seed, gb, hb);
    // ---- query_lstm_layernorm + dropout
    float mean, rstd;
    ln_stats(hb, Q, a.eps, red, mean, rstd);
    if (threadIdx.x == 0) { a.st_q[2 * b] = mean; a.st_q[2 * b + 1] = rstd; }
    for (int j = threadIdx.x; j < Q; j += DT) {
        float v = (hb[j] - mean) * rstd * a.lnq_g[j] + a.lnq_b[j];
        if (a.p_q > 0.f) v *= drop_scale(seed, a.site_q, (uint64_t)b * Q + j, a.p_q);
        a.qcur[(int64_t)b * Q + j] = v;
        qb[j] = v;
    }
    __syncthreads();
    // ---- attention scores of every stream: score_p = K'[b,p,:] . q * scale  (one wave per dot)
    for (int d = w; d < ns * P; d += DT / 64) {
        const int s = d / P, p = d % P;
        const float* kp = a.Kp[s] + ((int64_t)bk * P + p) * Q;
        float acc = 0.f;
#pragma unroll 4
        for (int j = lane * V; j < Q; j += 64 * V) {
            float kv[V];
            vload<V>(kv, kp + j);
#pragma unroll
            for (int e = 0; e < V; ++e) acc += kv[e] * qb[j + e];
        }
        acc = wave_sum(acc);
        if (lane == 0) sc[s * MAXP + p] = acc * a.scale;
    }
    __syncthreads();
    if (threadIdx.x < ns * P) {
        const int s = threadIdx.x / P, p = threadIdx.x % P;
        float m = -INFINITY;
        for (int i = 0; i < P; ++i) m = fmaxf(m, sc[s * MAXP + i]);
        float l = 0.f;
        for (int i = 0; i < P; ++i) l += __expf(sc[s * MAXP + i] - m);
        const float wv = __expf(sc[s * MAXP + p] - m) / l;
        wt[s * MAXP + p] = wv;
        a.alpha[(int64_t)b * ns * P + s * P + p] = wv;
    }
    __syncthreads();
    // ---- context_s = sum_p w_p V'[b,p,:] for both streams; tanh(context) stays in LDS for the LayerNorms
    float* cb = gb;
    for (int it = threadIdx.x * V; it < ns * H; it += DT * V) {
        const int s = it / H, j = it % H;
        const float* vp = a.Vp[s] + (int64_t)bk * P * H + j;
        float acc[V], t[4][V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        int p = 0;
        for (; p + 4 <= P; p += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) vload<V>(t[u], vp + (int64_t)(p + u) * H);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < V; ++e) acc[e] += wt[s * MAXP + p + u] * t[u][e];
        }
        for (; p < P; ++p) {
            vload<V>(t[0], vp + (int64_t)p * H);
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] += wt[s * MAXP + p] * t[0][e];
        }
        vstore<V>(a.cpre[s] + (int64_t)b * H + j, acc);
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = tanhf(acc[e]);
        vstore<V>(cb + s * MAXW + j, acc);
    }
    __syncthreads();
    // ---- output_layer LayerNorm (+dropout) of both streams, statistics reduced together
    float s0 = 0.f, s1 = 0.f;
    for (int j = threadIdx.x; j < H; j += DT) { s0 += cb[j]; if (ns > 1) s1 += cb[MAXW + j]; }
    block_sum2(s0, s1, red);
    const float m0 = s0 / H, m1 = s1 / H;
    float q0 = 0.f, q1 = 0.f;
    for (int j = threadIdx.x; j < H; j += DT) {
        const float d0 = cb[j] - m0; q0 += d0 * d0;
        if (ns > 1) { const float d1 = cb[MAXW + j] - m1; q1 += d1 * d1; }
    }
    block_sum2(q0, q1, red);
    const float r0 = rsqrtf(q0 / H + a.eps), r1 = rsqrtf(q1 / H + a.eps);
    if (threadIdx.x == 0) {
        a.st_c[0][2 * b] = m0; a.st_c[0][2 * b + 1] = r0;
        if (ns > 1) { a.st_c[1][2 * b] = m1; a.st_c[1][2 * b + 1] = r1; }
    }
    for (int it = threadIdx.x; it < ns * H; it += DT) {
        const int s = it / H, j = it % H;
        float v = (cb[s * MAXW + j] - (s ? m1 : m0)) * (s ? r1 : r0) * a.lnc_g[s][j] + a.lnc_b[s][j];
        if (a.p_att[s] > 0.f) v *= drop_scale(seed, a.site_att[s], (uint64_t)b * H + j, a.p_att[s]);
        a.ctx[s][(int64_t)b * H + j] = v;
    }
}

template <int V>
__global__ __launch_bounds__(DT) void dec_tail_fwd_kernel(const dlsg_dec_tail_args a) {
    __shared__ __attribute__((aligned(16))) float gb[4 * MAXW];
    __shared__ __attribute__((aligned(16))) float hb[MAXW];
    __shared__ float red[32];
    const int b = blockIdx.x;
    const int D = a.D;
    const uint64_t seed = a.seed + (a.seed_ptr ? *a.seed_ptr : 0ull);
    constexpr int ND = MAXW / DT;
    float lg[ND], lb[ND];                     // (requested with the gates, used after the statistics)
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const int j = threadIdx.x + i * DT;
        lg[i] = j < D ? a.ln_g[j] : 0.f; lb[i] = j < D ? a.ln_b[j] : 0.f;
    }
    cell_row<V>(b, D, a.slabs, a.nslab, a.slab_stride, nullptr, 0, a.b_ih, a.b_hh, a.c_prev, a.c, a.hd, a.gates, a.p, a.site,
                seed, gb, hb);
    float mean, rstd;
    ln_stats(hb, D, a.eps, red, mean, rstd);
    if (threadIdx.x == 0) { a.st_l[2 * b] = mean; a.st_l[2 * b + 1] = rstd; }
    const bool sample = a.s_coins != nullptr && a.s_coins[a.s_t] == 0;        // block-uniform
    float dv[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const int j = threadIdx.x + i * DT;
        dv[i] = 0.f;
        if (j >= D) break;
        dv[i] = tanhf((hb[j] - mean) * rstd * lg[i] + lb[i]);
        a.dout[(int64_t)b * D + j] = dv[i];
    }
    if (!sample) return;
    // ---- the next word is sampled from this row's own logits: project it (dout . s_W^T + s_b), first maximum, embed
    __syncthreads();                              // (hb was read by everyone above)
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const int j = threadIdx.x + i * DT;
        if (j < D) hb[j] = dv[i];
    }
    __syncthreads();
    __shared__ int s_bi[DT / 64];
    __shared__ int64_t s_chosen;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    constexpr int CU4 = 4;                        // vocabulary rows per wave and trip: their loads are in flight together
    for (int v0 = w * CU4; v0 < a.s_V; v0 += (DT / 64) * CU4) {
        float acc[CU4];
#pragma unroll
        for (int u = 0; u < CU4; ++u) {
            acc[u] = 0.f;
            const int v = min(v0 + u, a.s_V - 1);
            const float* wr = a.s_W + (int64_t)v * D;
            if (V == 4) {
                for (int j = 4 * lane; j < D; j += 256) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(hb + j);
                    const f32x4 y = *reinterpret_cast<const f32x4*>(wr + j);
                    acc[u] += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
                }
            } else {
                for (int j = lane; j < D; j += 64) acc[u] += hb[j] * wr[j];
            }
        }
#pragma unroll
        for (int u = 0; u < CU4; ++u) {
            const int v = v0 + u;
            float s_ = wave_sum(acc[u]);
            if (a.s_b) s_ += a.s_b[min(v, a.s_V - 1)];
            if (v < a.s_V && (s_ > best || (s_ == best && v < bidx))) { best = s_; bidx = v; }
        }
    }
    if (lane == 0) { red[w] = best; s_bi[w] = bidx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < DT / 64; ++k)
            if (red[k] > best || (red[k] == best && s_bi[k] < bidx)) { best = red[k]; bidx = s_bi[k]; }
        if (bidx == 0x7fffffff) bidx = 0;       // a row without a maximum (all NaN / -inf): word 0, never an out-of-range gather below
        s_chosen = bidx;
        a.s_ids[b] = bidx;
    }
    __syncthreads();
    const int64_t id = s_chosen;
    for (int j = threadIdx.x; j < a.s_Wd; j += DT) {
        float e = a.s_E[id * a.s_Wd + j];
        if (a.s_p > 0.f) e *= drop_scale(seed, a.s_site, (uint64_t)(a.s_row0 + b) * a.s_Wd + j, a.s_p);
        a.s_we[(int64_t)b * a.s_ldwe + j] = e;
    }
}


template <int N>
__device__ __forceinline__ void block_sumN(float (&v)[N], float* red) {   // red >= 16*N floats
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = wave_sum(v[i]);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < N; ++i) red[16 * i + w] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float r = 0.f;
#pragma unroll
        for (int k = 0; k < DT / 64; ++k) r += red[16 * i + k];
        v[i] = r;
    }
}

// Backward of dec_mid_fwd_kernel for one batch row.  LDS holds the summed input gradient, d(pre-tanh context) of
// both streams and the per-stream dq partials; LayerNorm temporaries live in registers across the reductions.
template <int V>
__global__ __launch_bounds__(DT) void dec_mid_bwd_kernel(const dlsg_dec_mid_bwd_args a) {
    __shared__ __attribute__((aligned(16))) float dxb[3 * MAXW];   // [d ctx_0 | d ctx_1 | d q_cur] (dense: ns*H + Q)
    __shared__ __attribute__((aligned(16))) float dcp[2 * MAXW];   // d(pre-tanh context), stream s at s*MAXW
    __shared__ __attribute__((aligned(16))) float dqb[2 * MAXW];   // dq contribution of stream s at s*MAXW
    __shared__ float dwb[2 * MAXP], dsb[2 * MAXP], wb[2 * MAXP];
    __shared__ float red[64];
    constexpr int IC = 2 * MAXW / DT, IQ = MAXW / DT;
    const int b = blockIdx.x;
    const int Q = a.Q, H = a.H, D = a.D, P = a.P, ns = a.nstream;
    const int NX = ns * H + Q, WT = NX + D;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t seed = a.seed + (a.seed_ptr ? *a.seed_ptr : 0ull);

    // ---- loads that depend on nothing computed here go first: they overlap the slab sum
    float yv[IC], xq[IQ], rec[IQ];
#pragma unroll
    for (int i = 0; i < IC; ++i) {
        const int it = threadIdx.x + i * DT;
        yv[i] = 0.f;
        if (it < ns * H) { const int s = it / H, j = it % H; yv[i] = a.cpre[s][(int64_t)b * H + j]; }
    }
#pragma unroll
    for (int i = 0; i < IQ; ++i) {
        const int j = threadIdx.x + i * DT;
        xq[i] = 0.f; rec[i] = 0.f;
        if (j < Q) {
            xq[i] = a.qh[(int64_t)b * Q + j];
            if (a.rec_slabs) {
                // four slabs per round trip (a load -> add per slab waits for each load in turn)
                const float* rp = a.rec_slabs + (int64_t)b * a.rec_ld + j;
                int k = 0;
                for (; k + 4 <= a.rec_nslab; k += 4) {
                    const float t0 = rp[(int64_t)k * a.rec_slab_stride], t1 = rp[(int64_t)(k + 1) * a.rec_slab_stride],
                                t2 = rp[(int64_t)(k + 2) * a.rec_slab_stride], t3 = rp[(int64_t)(k + 3) * a.rec_slab_stride];
                    rec[i] += t0; rec[i] += t1; rec[i] += t2; rec[i] += t3;
                }
                for (; k < a.rec_nslab; ++k) rec[i] += rp[(int64_t)k * a.rec_slab_stride];
            }
        }
    }
    if (threadIdx.x < ns * P) {
        const int s = threadIdx.x / P, p = threadIdx.x % P;
        wb[s * MAXP + p] = a.alpha[(int64_t)b * ns * P + threadIdx.x];
    }
    // ---- sum the slabs of the language cell's input-gradient GEMM
    const int ncol = a.write_rec ? WT : NX;
    for (int col = threadIdx.x * V; col < ncol; col += DT * V) {
        float acc[V], t[4][V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        const float* sp = a.slabs + (int64_t)b * WT + col;
        int k = 0;
        for (; k + 4 <= a.nslab; k += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) vload<V>(t[u], sp + (k + u) * a.slab_stride);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < V; ++e) acc[e] += t[u][e];
        }
        if (k < a.nslab) {
            const int rem = a.nslab - k;
#pragma unroll
            for (int u = 0; u < 3; ++u)
                if (u < rem) vload<V>(t[u], sp + (k + u) * a.slab_stride);
#pragma unroll
            for (int u = 0; u < 3; ++u)
                if (u < rem)
#pragma unroll
                    for (int e = 0; e < V; ++e) acc[e] += t[u][e];
        }
        if (col < NX) vstore<V>(dxb + col, acc);
        else vstore<V>(a.dlh_rec + (int64_t)b * D + (col - NX), acc);
    }
    __syncthreads();
    // ---- output_layer LayerNorm backward of both streams (x = tanh(cpre))
    float xh[IC], gx[IC];
    float sums[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < IC; ++i) {
        const int it = threadIdx.x + i * DT;
        xh[i] = 0.f; gx[i] = 0.f;
        if (it < ns * H) {
            const int s = it / H, j = it % H;
            const float y = tanhf(yv[i]);
            yv[i] = y;
            const float mean = a.st_c[s][2 * b], rstd = a.st_c[s][2 * b + 1];
            xh[i] = (y - mean) * rstd;
            float g = dxb[it];
            if (a.p_att[s] > 0.f) g *= drop_scale(seed, a.site_att[s], (uint64_t)b * H + j, a.p_att[s]);
            float* pc = a.part_c[s] + (int64_t)b * 2 * H;
            pc[j] = g * xh[i];
            pc[H + j] = g;
            gx[i] = g * a.lnc_g[s][j];
            if (s == 0) { sums[0] += gx[i]; sums[1] += gx[i] * xh[i]; }
            else { sums[2] += gx[i]; sums[3] += gx[i] * xh[i]; }
        }
    }
    block_sumN<4>(sums, red);
#pragma unroll
    for (int i = 0; i < IC; ++i) {
        const int it = threadIdx.x + i * DT;
        if (it < ns * H) {
            const int s = it / H, j = it % H;
            const float rstd = a.st_c[s][2 * b + 1];
            const float m1 = sums[2 * s] / H, m2 = sums[2 * s + 1] / H;
            const float d = rstd * (gx[i] - m1 - xh[i] * m2) * (1.f - yv[i] * yv[i]);
            dcp[s * MAXW + j] = d;
            a.dcpre[s][(int64_t)b * H + j] = d;
        }
    }
    __syncthreads();
    // ---- attention backward: dw_p = V'[b,p,:] . dcpre (+ dalpha), softmax backward, dq
    for (int d = w; d < ns * P; d += DT / 64) {
        const int s = d / P, p = d % P;
        const float* vp = a.Vp[s] + ((int64_t)b * P + p) * H;
        float acc = 0.f;
#pragma unroll 4
        for (int j = lane * V; j < H; j += 64 * V) {
            float vv[V];
            vload<V>(vv, vp + j);
#pragma unroll
            for (int e = 0; e < V; ++e) acc += vv[e] * dcp[s * MAXW + j + e];
        }
        acc = wave_sum(acc);
        if (lane == 0) dwb[s * MAXP + p] = acc + (a.dalpha ? a.dalpha[(int64_t)b * ns * P + s * P + p] : 0.f);
    }
    __syncthreads();
    if (threadIdx.x < ns * P) {
        const int s = threadIdx.x / P, p = threadIdx.x % P;
        float dot = 0.f;
        for (int i = 0; i < P; ++i) dot += wb[s * MAXP + i] * dwb[s * MAXP + i];
        const float v = wb[s * MAXP + p] * (dwb[s * MAXP + p] - dot) * a.scale;
        dsb[s * MAXP + p] = v;
        a.ds[(int64_t)b * ns * P + s * P + p] = v;
    }
    __syncthreads();
    for (int it = threadIdx.x * V; it < ns * Q; it += DT * V) {
        const int s = it / Q, j = it % Q;
        const float* kp = a.Kp[s] + (int64_t)b * P * Q + j;
        float acc[V], t[4][V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        int p = 0;
        for (; p + 4 <= P; p += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) vload<V>(t[u], kp + (int64_t)(p + u) * Q);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < V; ++e) acc[e] += dsb[s * MAXP + p + u] * t[u][e];
        }
        for (; p < P; ++p) {
            vload<V>(t[0], kp + (int64_t)p * Q);
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] += dsb[s * MAXP + p] * t[0][e];
        }
        vstore<V>(dqb + s * MAXW + j, acc);
    }
    __syncthreads();
    // ---- query_lstm_layernorm backward + recurrent part + query cell backward
    float gq[IQ], xhq[IQ];
    float s2[2] = {0.f, 0.f};
    const float meanq = a.st_q[2 * b], rstdq = a.st_q[2 * b + 1];
#pragma unroll
    for (int i = 0; i < IQ; ++i) {
        const int j = threadIdx.x + i * DT;
        gq[i] = 0.f; xhq[i] = 0.f;
        if (j < Q) {
            float g = dxb[ns * H + j] + dqb[j];
            if (ns > 1) g += dqb[MAXW + j];
            if (a.p_q > 0.f) g *= drop_scale(seed, a.site_q, (uint64_t)b * Q + j, a.p_q);
            xhq[i] = (xq[i] - meanq) * rstdq;
            float* pq = a.part_q + (int64_t)b * 2 * Q;
            pq[j] = g * xhq[i];
            pq[Q + j] = g;
            gq[i] = g * a.lnq_g[j];
            s2[0] += gq[i]; s2[1] += gq[i] * xhq[i];
        }
    }
    block_sumN<2>(s2, red);
    const float m1 = s2[0] / Q, m2 = s2[1] / Q;
#pragma unroll
    for (int i = 0; i < IQ; ++i) {
        const int j = threadIdx.x + i * DT;
        if (j < Q) {
            const float dh = rstdq * (gq[i] - m1 - xhq[i] * m2) + rec[i];
            const float* gp = a.gates + (int64_t)b * 4 * Q;
            const float ig = gp[j], fg = gp[Q + j], gg = gp[2 * Q + j], og = gp[3 * Q + j];
            const float c = a.c[(int64_t)b * Q + j];
            const float cp = a.c_prev ? a.c_prev[(int64_t)b * Q + j] : 0.f;
            const float tc = tanhf(c);
            const float dc = dh * og * (1.f - tc * tc) + a.dc[(int64_t)b * Q + j];
            float* dgp = a.dgates + (int64_t)b * 4 * Q;
            dgp[j] = dc * gg * ig * (1.f - ig);
            dgp[Q + j] = dc * cp * fg * (1.f - fg);
            dgp[2 * Q + j] = dc * ig * (1.f - gg * gg);
            dgp[3 * Q + j] = dh * tc * og * (1.f - og);
            a.dc[(int64_t)b * Q + j] = dc * fg;
        }
    }
}

// dK' / dV' of the attention caches: a (P x L)(L x N) contraction per batch row, L = number of word steps.
// grid (B, 2*nstream): blockIdx.y = 2*s + kind (0: dV' from alpha & dcpre, 1: dK' from ds & q_cur).
constexpr int CG_PC = 8;
template <int V>
__global__ __launch_bounds__(256) void decatt_cache_grads_kernel(const dlsg_decatt_cache_grads_args a) {
    const int b = blockIdx.x, s = blockIdx.y >> 1, kind = blockIdx.y & 1;
    const int N = kind ? a.Q : a.H, P = a.P, nsP = a.nstream * a.P;
    const float* coef = (kind ? a.ds : a.alpha) + (int64_t)b * nsP + s * P;     // + t*B*nsP
    const float* src = (kind ? a.qcur : a.dcpre[s]) + (int64_t)b * N;            // + t*B*N
    float* dst = (kind ? a.dKp[s] : a.dVp[s]) + (int64_t)b * P * N;
    for (int p0 = 0; p0 < P; p0 += CG_PC)
        for (int j = threadIdx.x * V; j < N; j += 256 * V) {
            float acc[CG_PC][V];
#pragma unroll
            for (int p = 0; p < CG_PC; ++p)
#pragma unroll
                for (int e = 0; e < V; ++e) acc[p][e] = 0.f;
#pragma unroll 2
            for (int t = 0; t < a.L; ++t) {
                float x[V];
                vload<V>(x, src + (int64_t)t * a.B * N + j);
                const float* cf = coef + (int64_t)t * a.B * nsP + p0;
#pragma unroll
                for (int p = 0; p < CG_PC; ++p) {
                    const float cv = (p0 + p < P) ? cf[p] : 0.f;
#pragma unroll
                    for (int e = 0; e < V; ++e) acc[p][e] += cv * x[e];
                }
            }
#pragma unroll
            for (int p = 0; p < CG_PC; ++p)
                if (p0 + p < P) vstore<V>(dst + (int64_t)(p0 + p) * N + j, acc[p]);
        }
}

bool vec_ok(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int dlsg_dec_mid_fwd(const dlsg_dec_mid_args* a, void* stream) {
    if (!a || a->Q < 1 || a->Q > MAXW || a->H < 1 || a->H > MAXW || a->P < 1 || a->P > MAXP || a->nstream < 1 || a->nstream > 2)
        return DLSG_EINVAL;
    if (a->B == 0) return DLSG_OK;
    bool v4 = a->Q % 4 == 0 && a->H % 4 == 0 && a->slab_stride % 4 == 0 && a->ldadd % 4 == 0 && vec_ok(a->slabs) &&
              vec_ok(a->addend) && vec_ok(a->b_ih) && vec_ok(a->b_hh) && vec_ok(a->gates);
    for (int s = 0; s < a->nstream; ++s) v4 = v4 && vec_ok(a->Kp[s]) && vec_ok(a->Vp[s]) && vec_ok(a->cpre[s]);
    hipLaunchKernelGGL(v4 ? dec_mid_fwd_kernel<4> : dec_mid_fwd_kernel<1>, dim3(a->B), dim3(DT), 0,
                       reinterpret_cast<hipStream_t>(stream), *a);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_dec_tail_fwd(const dlsg_dec_tail_args* a, void* stream) {
    if (!a || a->D < 1 || a->D > MAXW) return DLSG_EINVAL;
    if (a->B == 0) return DLSG_OK;
    if (a->s_coins && (!a->s_W || !a->s_E || !a->s_ids || !a->s_we || a->s_V < 1 || a->s_Wd < 1)) return DLSG_EINVAL;
    const bool v4 = a->D % 4 == 0 && a->slab_stride % 4 == 0 && vec_ok(a->slabs) && vec_ok(a->b_ih) && vec_ok(a->b_hh) &&
                    vec_ok(a->gates) && (!a->s_coins || vec_ok(a->s_W));
    hipLaunchKernelGGL(v4 ? dec_tail_fwd_kernel<4> : dec_tail_fwd_kernel<1>, dim3(a->B), dim3(DT), 0,
                       reinterpret_cast<hipStream_t>(stream), *a);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_dec_mid_bwd(const dlsg_dec_mid_bwd_args* a, void* stream) {
    if (!a || a->Q < 1 || a->Q > MAXW || a->H < 1 || a->H > MAXW || a->D < 0 || a->P < 1 || a->P > MAXP || a->nstream < 1 ||
        a->nstream > 2)
        return DLSG_EINVAL;
    if (a->B == 0) return DLSG_OK;
    bool v4 = a->Q % 4 == 0 && a->H % 4 == 0 && a->D % 4 == 0 && a->slab_stride % 4 == 0 && vec_ok(a->slabs) && vec_ok(a->dlh_rec);
    for (int s = 0; s < a->nstream; ++s) v4 = v4 && vec_ok(a->Kp[s]) && vec_ok(a->Vp[s]);
    hipLaunchKernelGGL(v4 ? dec_mid_bwd_kernel<4> : dec_mid_bwd_kernel<1>, dim3(a->B), dim3(DT), 0,
                       reinterpret_cast<hipStream_t>(stream), *a);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_decatt_cache_grads(const dlsg_decatt_cache_grads_args* a, void* stream) {
    if (!a || a->L < 0 || a->P < 1 || a->nstream < 1 || a->nstream > 2) return DLSG_EINVAL;
    if (a->B == 0) return DLSG_OK;
    bool v4 = a->Q % 4 == 0 && a->H % 4 == 0 && vec_ok(a->qcur);
    for (int s = 0; s < a->nstream; ++s) v4 = v4 && vec_ok(a->dcpre[s]) && vec_ok(a->dKp[s]) && vec_ok(a->dVp[s]);
    hipLaunchKernelGGL(v4 ? decatt_cache_grads_kernel<4> : decatt_cache_grads_kernel<1>, dim3(a->B, 2 * a->nstream), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), *a);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
