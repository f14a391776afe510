// Row-wise / pointwise kernels of the D-LSG hot path (gfx950): tanh+LayerNorm(+PE)(+dropout) forward/backward,
// strided softmax, LSTM cell pointwise, embedding gather/scatter, argmax, ragged cross entropy, Adam.
// All are HBM/L2-bound streaming kernels: one pass over the row held in registers, float4 where aligned.
#include "common.hpp"
#include "dlsg.h"

using namespace dlsg;

namespace {

constexpr int LN_THREADS = 256;
constexpr int LN_MAXPT = 8;  // n <= 2048

// ------------------------------------------------------------------------------------------------ rowln fwd
// (several norms of one shape per launch -- the two encoder streams' obj_visual_norm, visual_norm: blockIdx.y picks the block)
struct RowLnPack { dlsg_rowln_args s[DLSG_ROWLN_MAXMULTI]; };
struct RowLnBwdPack { dlsg_rowln_bwd_args s[DLSG_ROWLN_MAXMULTI]; };

__global__ __launch_bounds__(LN_THREADS) void rowln_fwd_kernel(const RowLnPack pk) {
    __shared__ float red[16];
    const dlsg_rowln_args& a = pk.s[blockIdx.y];
    const int n = a.n;
    const uint64_t seed = a.seed + (a.seed_ptr ? *a.seed_ptr : 0ull);
    // gamma / beta are the same for every row, the positional row is known up front: requested with the row itself, not a
    // memory round trip after the two reductions
    float gam[LN_MAXPT], bet[LN_MAXPT];
#pragma unroll
    for (int i = 0; i < LN_MAXPT; ++i) {
        const int j = threadIdx.x + i * LN_THREADS;
        gam[i] = j < n ? a.gamma[j] : 0.f; bet[i] = j < n ? a.beta[j] : 0.f;
    }
    for (int row = blockIdx.x; row < a.rows; row += gridDim.x) {
        const float* x = a.x + (int64_t)row * a.ldx;
        const float* res = a.res ? a.res + (int64_t)row * a.ldres : nullptr;
        const float* pe = a.pe ? a.pe + (int64_t)(row % a.pe_rows) * n : nullptr;
        float t[LN_MAXPT], pev[LN_MAXPT];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXPT; ++i) {
            const int j = threadIdx.x + i * LN_THREADS;
            float z = 0.f;
            pev[i] = (pe && j < n) ? pe[j] : 0.f;
            if (j < n) {
                z = x[j];
                if (res) z += res[j];
                if (a.pre_tanh == 1) z = tanhf(z);
                s += z;
            }
            t[i] = z;
        }
        const float mean = block_sum(s, red) / n;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXPT; ++i) {
            const int j = threadIdx.x + i * LN_THREADS;
            if (j < n) { const float d = t[i] - mean; q += d * d; }
        }
        const float rstd = rsqrtf(block_sum(q, red) / n + a.eps);
        if (a.stats && threadIdx.x == 0) { a.stats[2 * (int64_t)row] = mean; a.stats[2 * (int64_t)row + 1] = rstd; }
        float* y = a.y + (int64_t)row * a.ldy;
#pragma unroll
        for (int i = 0; i < LN_MAXPT; ++i) {
            const int j = threadIdx.x + i * LN_THREADS;
            if (j < n) {
                float v = (t[i] - mean) * rstd * gam[i] + bet[i];
                if (a.post_tanh) v = tanhf(v);
                const uint64_t idx = (uint64_t)row * n + j;
                if (a.p1 > 0.f) v *= drop_scale(seed, a.site1, idx, a.p1);
                if (pe) {
                    v += pev[i];
                    if (a.p2 > 0.f) v *= drop_scale(seed, a.site2, idx, a.p2);
                }
                y[j] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ rowln bwd
// Each block walks rows blockIdx.x, +gridDim.x, ... and keeps per-column dgamma/dbeta partial sums in registers.
__global__ __launch_bounds__(LN_THREADS) void rowln_bwd_kernel(const RowLnBwdPack pk) {
    __shared__ float red[16];
    const dlsg_rowln_bwd_args& b = pk.s[blockIdx.y];
    const dlsg_rowln_args& a = b.f;
    const int n = a.n;
    const uint64_t seed = a.seed + (a.seed_ptr ? *a.seed_ptr : 0ull);
    float dg[LN_MAXPT], db[LN_MAXPT];
#pragma unroll
    for (int i = 0; i < LN_MAXPT; ++i) { dg[i] = 0.f; db[i] = 0.f; }
    for (int row = blockIdx.x; row < a.rows; row += gridDim.x) {
        const float* x = a.x + (int64_t)row * a.ldx;
        const float* res = a.res ? a.res + (int64_t)row * a.ldres : nullptr;
        const float* dy = b.dy + (int64_t)row * b.lddy;
        float* dx = b.dx + (int64_t)row * b.lddx;
        float mean, rstd;
        float t[LN_MAXPT], dxo[LN_MAXPT];
#pragma unroll
        for (int i = 0; i < LN_MAXPT; ++i) {       // dx += : the old values are requested with the row, not after the reductions
            const int j = threadIdx.x + i * LN_THREADS;
            dxo[i] = (b.accum_dx && j < n) ? dx[j] : 0.f;
        }
        if (a.stats) {
            mean = a.stats[2 * (int64_t)row]; rstd = a.stats[2 * (int64_t)row + 1];
#pragma unroll
            for (int i = 0; i < LN_MAXPT; ++i) {
                const int j = threadIdx.x + i * LN_THREADS;
                float z = 0.f;
                if (j < n) { z = x[j]; if (res) z += res[j]; if (a.pre_tanh == 1) z = tanhf(z); }
                t[i] = z;
            }
        } else {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < LN_MAXPT; ++i) {
                const int j = threadIdx.x + i * LN_THREADS;
                float z = 0.f;
                if (j < n) { z = x[j]; if (res) z += res[j]; if (a.pre_tanh == 1) z = tanhf(z); s += z; }
                t[i] = z;
            }
            mean = block_sum(s, red) / n;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < LN_MAXPT; ++i) {
                const int j = threadIdx.x + i * LN_THREADS;
                if (j < n) { const float d = t[i] - mean; q += d * d; }
            }
            rstd = rsqrtf(block_sum(q, red) / n + a.eps);
        }
        float gx[LN_MAXPT];   // dL/dxhat
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXPT; ++i) {
            const int j = threadIdx.x + i * LN_THREADS;
            gx[i] = 0.f;
            if (j < n) {
                const float xh = (t[i] - mean) * rstd;
                float g = dy[j];
                const uint64_t idx = (uint64_t)row * n + j;
                if (a.pe && a.p2 > 0.f) g *= drop_scale(seed, a.site2, idx, a.p2);
                if (a.p1 > 0.f) g *= drop_scale(seed, a.site1, idx, a.p1);
                if (a.post_tanh) {
                    const float yp = tanhf(xh * a.gamma[j] + a.beta[j]);
                    g *= (1.f - yp * yp);
                }
                dg[i] += g * xh;
                db[i] += g;
                const float gxh = g * a.gamma[j];
                gx[i] = gxh;
                s1 += gxh;
                s2 += gxh * xh;
            }
        }
        const float m1 = block_sum(s1, red) / n;
        const float m2 = block_sum(s2, red) / n;
#pragma unroll
        for (int i = 0; i < LN_MAXPT; ++i) {
            const int j = threadIdx.x + i * LN_THREADS;
            if (j < n) {
                const float xh = (t[i] - mean) * rstd;
                float d = rstd * (gx[i] - m1 - xh * m2);
                if (a.pre_tanh) d *= (1.f - t[i] * t[i]);   // mode 1: t = tanh(x); mode 2: x is already a tanh output
                dx[j] = d + dxo[i];
            }
        }
    }
    if (b.dgb_part) {
        float* pg = b.dgb_part + (int64_t)blockIdx.x * 2 * n;
#pragma unroll
        for (int i = 0; i < LN_MAXPT; ++i) {
            const int j = threadIdx.x + i * LN_THREADS;
            if (j < n) { pg[j] = dg[i]; pg[n + j] = db[i]; }
        }
    }
}

// ------------------------------------------------------------------------------------------------ column sum
// block = 64 columns x 16 row-lanes over one row chunk (gridDim.y chunks); fixed-order tree through LDS inside a chunk.
// One chunk: plain store (deterministic).  Several chunks (tall inputs, e.g. the 26624-row bias gradients): the
// per-chunk sums are combined with float atomics into `out`, which the caller has initialised (accum semantics).
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ part, int64_t ld, int rows, int n,
                                                      float* __restrict__ out, int accum, int rows_per_chunk,
                                                      float* __restrict__ ws) {
    __shared__ float red[16][64];
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + c;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    float s = 0.f;
    if (col < n)
        for (int r = r0 + rl; r < r1; r += 16) s += part[(int64_t)r * ld + col];
    red[rl][c] = s;
    __syncthreads();
    if (rl == 0 && col < n) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][c];
        if (gridDim.y > 1 && ws) {
            ws[(int64_t)blockIdx.y * n + col] = t;            // chunk partial: colsum_finish_kernel adds the chunks in order
        } else if (gridDim.y > 1) {
            atomicAdd(out + col, t);
        } else {
            if (accum) t += out[col];
            out[col] = t;
        }
    }
}

// 16-byte variant for the common case (n, ld multiples of 4, aligned base): block = 16 column quads x 64 row-lanes, four
// independent float4 loads in flight per thread (the scalar kernel above runs 64 workgroups of dependent adds on a
// 1664 x 4096 bias gradient: 0.6 TB/s).  Two destinations: `mode` 0 -> columns [0,split) go to out, [split,n) to out2
// (the gamma | beta halves of the LayerNorm partials); mode 1 -> every column goes to both (bias_ih and bias_hh of an
// LSTM receive the same gradient).
__global__ __launch_bounds__(1024) void colsum_v4_kernel(const float* __restrict__ part, int64_t ld, int rows, int n,
                                                         float* __restrict__ out, float* __restrict__ out2, int split, int mode,
                                                         int accum, int rows_per_chunk, float* __restrict__ ws) {
    __shared__ float red[64][65];
    const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int col = blockIdx.x * 64 + cg * 4;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (col < n) {
        int r = r0 + rl;
        for (; r + 192 < r1; r += 256) {
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(part + (int64_t)r * ld + col);
            const f32x4 t1 = *reinterpret_cast<const f32x4*>(part + (int64_t)(r + 64) * ld + col);
            const f32x4 t2 = *reinterpret_cast<const f32x4*>(part + (int64_t)(r + 128) * ld + col);
            const f32x4 t3 = *reinterpret_cast<const f32x4*>(part + (int64_t)(r + 192) * ld + col);
            acc += (t0 + t1) + (t2 + t3);
        }
        for (; r < r1; r += 64) acc += *reinterpret_cast<const f32x4*>(part + (int64_t)r * ld + col);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[rl][cg * 4 + e] = acc[e];
    __syncthreads();
    // 64 columns x 16 partial lanes, then a 16-wide shuffle tree
    const int c = threadIdx.x >> 4, part16 = threadIdx.x & 15;
    float t = red[part16][c] + red[part16 + 16][c] + red[part16 + 32][c] + red[part16 + 48][c];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    const int oc = blockIdx.x * 64 + c;
    if (part16 == 0 && oc < n) {
        float* d0 = (mode == 0 && oc >= split) ? out2 + (oc - split) : out + oc;
        float* d1 = mode == 1 ? out2 + oc : nullptr;
        if (gridDim.y > 1 && ws) {
            ws[(int64_t)blockIdx.y * n + oc] = t;
        } else if (gridDim.y > 1) {
            atomicAdd(d0, t);
            if (d1) atomicAdd(d1, t);
        } else {
            *d0 = accum ? *d0 + t : t;
            if (d1) *d1 = accum ? *d1 + t : t;
        }
    }
}

// Several short column sums in ONE launch (blockIdx.y picks the descriptor): the backward of a train step folds ~25 small
// partial arrays (LayerNorm dgamma | dbeta partials, bias gradients) of a few hundred to a few thousand rows each -- 8-9 us per
// launch on their own, almost all of it launch floor.  Same arithmetic and order of additions as colsum_v4_kernel with one chunk.
struct ColsumPack {
    dlsg_colsum_desc d[DLSG_COLSUM_MAXMULTI];
};
__global__ __launch_bounds__(1024) void colsum_multi_kernel(const ColsumPack pk) {
    __shared__ float red[64][65];
    const dlsg_colsum_desc& a = pk.d[blockIdx.y];
    if ((int)blockIdx.x * 64 >= a.n) return;
    const float* __restrict__ part = a.part;
    const int64_t ld = a.ld;
    const int rows = a.rows, n = a.n;
    const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int col = blockIdx.x * 64 + cg * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (col < n) {
        int r = rl;
        for (; r + 192 < rows; r += 256) {
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(part + (int64_t)r * ld + col);
            const f32x4 t1 = *reinterpret_cast<const f32x4*>(part + (int64_t)(r + 64) * ld + col);
            const f32x4 t2 = *reinterpret_cast<const f32x4*>(part + (int64_t)(r + 128) * ld + col);
            const f32x4 t3 = *reinterpret_cast<const f32x4*>(part + (int64_t)(r + 192) * ld + col);
            acc += (t0 + t1) + (t2 + t3);
        }
        for (; r < rows; r += 64) acc += *reinterpret_cast<const f32x4*>(part + (int64_t)r * ld + col);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[rl][cg * 4 + e] = acc[e];
    __syncthreads();
    const int c = threadIdx.x >> 4, part16 = threadIdx.x & 15;
    float t = red[part16][c] + red[part16 + 16][c] + red[part16 + 32][c] + red[part16 + 48][c];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    const int oc = blockIdx.x * 64 + c;
    if (part16 == 0 && oc < n) {
        const int mode = a.out_b ? (a.dup ? 1 : 0) : 0;
        const int split = a.out_b ? (a.dup ? n : a.split) : n;
        float* d0 = (mode == 0 && oc >= split) ? a.out_b + (oc - split) : a.out_a + oc;
        float* d1 = mode == 1 ? a.out_b + oc : nullptr;
        *d0 = a.accum ? *d0 + t : t;
        if (d1) *d1 = a.accum ? *d1 + t : t;
    }
}

// chunk partials ws (chunks, n) -> destinations, chunks added in order (bit-reproducible, unlike the atomic combine)
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ ws, int chunks, int n, float* __restrict__ out,
                                                            float* __restrict__ out2, int split, int mode, int accum) {
    const int oc = blockIdx.x * 256 + threadIdx.x;
    if (oc >= n) return;
    float t = 0.f;
    for (int c = 0; c < chunks; ++c) t += ws[(int64_t)c * n + oc];
    float* d0 = (mode == 0 && out2 && oc >= split) ? out2 + (oc - split) : out + oc;
    *d0 = accum ? *d0 + t : t;
    if (mode == 1 && out2) out2[oc] = accum ? out2[oc] + t : t;
}

// ------------------------------------------------------------------------------------------------ softmax (outer, n, inner)
// one wave per (outer, inner) line; lanes stride over n.
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                                          float* __restrict__ y, int64_t outer, int n, int inner) {
    const int lane = threadIdx.x & 63;
    const int64_t line = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (line >= outer * inner) return;
    const int64_t o = line / inner, in = line % inner;
    const float* xp = x + o * n * inner + in;
    const float* mp = mask ? mask + o * n * inner + in : nullptr;
    float* yp = y + o * n * inner + in;
    float m = -INFINITY;
    for (int j = lane; j < n; j += 64) {
        float v = xp[(int64_t)j * inner];
        if (mp && !(mp[(int64_t)j * inner] > 0.f)) v = -9e15f;
        m = fmaxf(m, v);
    }
    m = wave_max(m);
    float s = 0.f;
    for (int j = lane; j < n; j += 64) {
        float v = xp[(int64_t)j * inner];
        if (mp && !(mp[(int64_t)j * inner] > 0.f)) v = -9e15f;
        s += __expf(v - m);
    }
    s = wave_sum(s);
    const float inv = 1.f / s;
    for (int j = lane; j < n; j += 64) {
        float v = xp[(int64_t)j * inner];
        if (mp && !(mp[(int64_t)j * inner] > 0.f)) v = -9e15f;
        yp[(int64_t)j * inner] = __expf(v - m) * inv;
    }
}

__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                          float* __restrict__ dx, int64_t outer, int n, int inner) {
    const int lane = threadIdx.x & 63;
    const int64_t line = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (line >= outer * inner) return;
    const int64_t o = line / inner, in = line % inner;
    const int64_t base = o * n * inner + in;
    float s = 0.f;
    for (int j = lane; j < n; j += 64) s += y[base + (int64_t)j * inner] * dy[base + (int64_t)j * inner];
    s = wave_sum(s);
    for (int j = lane; j < n; j += 64) {
        const int64_t k = base + (int64_t)j * inner;
        dx[k] = y[k] * (dy[k] - s);
    }
}

// ------------------------------------------------------------------------------------------------ LSTM pointwise
// blockIdx.z selects one of up to two descriptors: the two directions of a BiLSTM step run as one launch.
struct PwFwdSet { dlsg_lstm_pw_args a[2]; };
struct PwBwdSet { dlsg_lstm_pw_bwd_args a[2]; };

__global__ __launch_bounds__(256) void lstm_pw_fwd_kernel(const PwFwdSet set) {
    const dlsg_lstm_pw_args& a = set.a[blockIdx.z];
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (j >= a.H) return;
    const int H = a.H;
    float pre[4] = {0.f, 0.f, 0.f, 0.f};
    // slab partial sums: all four gates of a slab pair are loaded before any add (independent loads in flight)
    const float* sp = a.slabs + (int64_t)b * 4 * H + j;
    int k = 0;
    for (; k + 2 <= a.nslab; k += 2) {
        float t[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int g = 0; g < 4; ++g) t[u][g] = sp[(k + u) * a.slab_stride + g * H];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int g = 0; g < 4; ++g) pre[g] += t[u][g];
    }
    if (k < a.nslab) {
#pragma unroll
        for (int g = 0; g < 4; ++g) pre[g] += sp[k * a.slab_stride + g * H];
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int col = g * H + j;
        float s = pre[g];
        if (a.addend) s += a.addend[(int64_t)b * a.ldadd + col];
        if (a.b_ih) s += a.b_ih[col];
        if (a.b_hh) s += a.b_hh[col];
        pre[g] = s;
    }
    const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf(pre[2]), og = sigmoidf_(pre[3]);
    const float cp = a.c_prev ? a.c_prev[(int64_t)b * a.ldcp + j] : 0.f;
    const float c = fg * cp + ig * gg;
    const float h = og * tanhf(c);
    a.c[(int64_t)b * a.ldc_ + j] = c;
    if (a.h) a.h[(int64_t)b * a.ldh + j] = h;
    if (a.h2) {
        float h2 = h;
        if (a.p > 0.f) h2 *= drop_scale(a.seed + (a.seed_ptr ? *a.seed_ptr : 0ull), a.site, (uint64_t)b * H + j, a.p);
        a.h2[(int64_t)b * a.ldh2 + j] = h2;
    }
    if (a.gates) {
        float* gp = a.gates + (int64_t)b * a.ldg;
        gp[j] = ig; gp[H + j] = fg; gp[2 * H + j] = gg; gp[3 * H + j] = og;
    }
}

__global__ __launch_bounds__(256) void lstm_pw_bwd_kernel(const PwBwdSet set) {
    const dlsg_lstm_pw_bwd_args& a = set.a[blockIdx.z];
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (j >= a.H) return;
    const int H = a.H;
    const float* gp = a.gates + (int64_t)b * a.ldg;
    const float ig = gp[j], fg = gp[H + j], gg = gp[2 * H + j], og = gp[3 * H + j];
    const float c = a.c[(int64_t)b * a.ldc_ + j];
    const float cp = a.c_prev ? a.c_prev[(int64_t)b * a.ldcp + j] : 0.f;
    float dh = a.dh ? a.dh[(int64_t)b * a.lddh + j] : 0.f;
    if (a.dh2 || a.dh3 || a.dh4) {
        float d2 = a.dh2 ? a.dh2[(int64_t)b * a.lddh2 + j] : 0.f;
        if (a.dh3) d2 += a.dh3[(int64_t)b * a.lddh3 + j];
        if (a.dh4) {
            const int ns4 = a.dh4_nslab > 1 ? a.dh4_nslab : 1;
            const float* hp = a.dh4 + (int64_t)b * a.lddh4 + j;
            int k = 0;
            for (; k + 4 <= ns4; k += 4) {             // four slabs per round trip; same order of additions
                const float t0 = hp[(int64_t)k * a.dh4_slab_stride], t1 = hp[(int64_t)(k + 1) * a.dh4_slab_stride],
                            t2 = hp[(int64_t)(k + 2) * a.dh4_slab_stride], t3 = hp[(int64_t)(k + 3) * a.dh4_slab_stride];
                d2 += t0; d2 += t1; d2 += t2; d2 += t3;
            }
            for (; k < ns4; ++k) d2 += hp[(int64_t)k * a.dh4_slab_stride];
        }
        if (a.p > 0.f) d2 *= drop_scale(a.seed + (a.seed_ptr ? *a.seed_ptr : 0ull), a.site, (uint64_t)b * H + j, a.p);
        dh += d2;
    }
    const float tc = tanhf(c);
    float dc = dh * og * (1.f - tc * tc);
    if (a.dc_next) dc += a.dc_next[(int64_t)b * a.lddcn + j];
    float* dgp = a.dgates + (int64_t)b * a.lddg;
    dgp[j] = dc * gg * ig * (1.f - ig);
    dgp[H + j] = dc * cp * fg * (1.f - fg);
    dgp[2 * H + j] = dc * ig * (1.f - gg * gg);
    dgp[3 * H + j] = dh * tc * og * (1.f - og);
    if (a.dc_prev) a.dc_prev[(int64_t)b * a.lddcp + j] = dc * fg;
}

// ------------------------------------------------------------------------------------------------ small movers
__global__ void mean_rows_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t ldo, int P, int H) {
    const int b = blockIdx.y;
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= H) return;
    float s = 0.f;
    for (int p = 0; p < P; ++p) s += x[((int64_t)b * P + p) * H + h];
    out[(int64_t)b * ldo + h] = s / P;
}
__global__ void mean_rows_bwd_kernel(const float* __restrict__ dout, int64_t lddo, float* __restrict__ dx, int P, int H,
                                     int accum) {
    const int b = blockIdx.y;
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= H) return;
    const float g = dout[(int64_t)b * lddo + h] / P;
    for (int p = 0; p < P; ++p) {
        float* d = dx + ((int64_t)b * P + p) * H + h;
        *d = accum ? *d + g : g;
    }
}

__global__ void embed_fwd_kernel(const float* __restrict__ E, const int64_t* __restrict__ ids, float* __restrict__ out,
                                 int64_t ldo, int W, float p, uint64_t seed, uint32_t site, int64_t row0, const uint64_t* seed_ptr) {
    if (seed_ptr) seed += *seed_ptr;
    const int r = blockIdx.x;
    const int64_t id = ids[r];
    for (int j = threadIdx.x; j < W; j += blockDim.x) {
        float v = E[id * W + j];
        if (p > 0.f) v *= drop_scale(seed, site, (uint64_t)(row0 + r) * W + j, p);
        out[(int64_t)r * ldo + j] = v;
    }
}
// dE[ids[r], :] += drop(dout[r, :]) without atomics, so that a step is bit-reproducible (float atomics add in arrival order:
// the word-embedding gradient differed in the last bit between runs of the same step).  One wave per row.  The first row that
// carries an id owns it and adds the rows with that id in row order; waves of later duplicates exit.  Matches are found 64 ids
// at a time with a ballot (eight independent id loads in flight), so only real matches cost a row read.
__global__ __launch_bounds__(1024) void embed_bwd_kernel(const float* __restrict__ dout, int64_t lddo, const int64_t* __restrict__ ids,
                                                         float* __restrict__ dE, int rows, int W, float p, uint64_t seed, uint32_t site,
                                                         int64_t row0, const uint64_t* seed_ptr) {
    if (seed_ptr) seed += *seed_ptr;
    // every wave of the workgroup finds the same matches (ids only) and owns 64 columns per pass: a frequent id (the <pad> word
    // of the caption tails: 40 % of the rows) is 650 rows added by ONE workgroup, and with the dropout hash per element that
    // was 87 us in a single wave
    const int r = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    const int64_t id = ids[r];
    // ---- an earlier row with the same id?  then that row's workgroup does the work
    for (int base = 0; base < r; base += 512) {
        int64_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + 64 * u + lane;
            v[u] = (i < r) ? ids[i] : -1;
        }
        bool hit = false;
#pragma unroll
        for (int u = 0; u < 8; ++u) hit = hit || (v[u] == id);
        if (__ballot(hit)) return;                       // wave-uniform, and the same in every wave
    }
    for (int c0 = 64 * wv; c0 < W; c0 += 64 * nwv) {       // this wave's column block of the pass
        const int j = c0 + lane;
        float acc = 0.f;
        for (int base = r; base < rows; base += 512) {
            int64_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + 64 * u + lane;
                v[u] = (i < rows) ? ids[i] : -1;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                unsigned long long m = __ballot(v[u] == id);
                while (m) {                               // ascending rows: the sum order is fixed
                    int rr[8];                            // up to eight matching rows fetched together, added in order
#pragma unroll
                    for (int x = 0; x < 8; ++x) {
                        rr[x] = m ? base + 64 * u + (__ffsll((long long)m) - 1) : -1;
                        m &= m - 1;
                    }
                    float g[8];
#pragma unroll
                    for (int x = 0; x < 8; ++x) g[x] = (rr[x] >= 0 && j < W) ? dout[(int64_t)rr[x] * lddo + j] : 0.f;
#pragma unroll
                    for (int x = 0; x < 8; ++x)
                        if (rr[x] >= 0 && j < W) {
                            float gv = g[x];
                            if (p > 0.f) gv *= drop_scale(seed, site, (uint64_t)(row0 + rr[x]) * W + j, p);
                            acc += gv;
                        }
                }
            }
        }
        if (j < W) dE[id * W + j] += acc;
    }
}

// first maximum wins (torch.max / argmax tie rule on CPU and GPU for exact ties: lowest index)
__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ logits, int64_t ld, int64_t* __restrict__ ids,
                                                     int V) {
    __shared__ float bv[4];
    __shared__ int bi[4];
    const int r = blockIdx.x;
    const float* x = logits + (int64_t)r * ld;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int j = threadIdx.x; j < V; j += blockDim.x) {
        const float v = x[j];
        if (v > best || (v == best && j < idx)) { best = v; idx = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { bv[w] = best; bi[w] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k)
            if (bv[k] > best || (bv[k] == best && bi[k] < idx)) { best = bv[k]; idx = bi[k]; }
        ids[r] = idx == 0x7fffffff ? 0 : idx;            // no maximum (a row of NaN / -inf): 0, as torch.argmax gives for -inf rows
    }
}

// scheduled sampling select + embedding gather: one block per batch row
__global__ __launch_bounds__(256) void select_embed_kernel(const float* __restrict__ logits, int64_t ld, int V,
                                                           const int64_t* __restrict__ captions, int L, int t,
                                                           const int32_t* __restrict__ coins, const float* __restrict__ E,
                                                           int64_t* __restrict__ ids_out, float* __restrict__ out,
                                                           int64_t ldo, int W, float p, uint64_t seed, uint32_t site,
                                                           int64_t row0, const uint64_t* seed_ptr, int prefilled) {
    __shared__ float bv[4];
    __shared__ int bi[4];
    __shared__ int64_t chosen;
    if (prefilled && coins[t] != 0) return;      // teacher-forced step whose word the caller embedded up front: nothing to do
    if (seed_ptr) seed += *seed_ptr;
    const int r = blockIdx.x;
    if (coins[t] != 0) {
        if (threadIdx.x == 0) chosen = captions[(int64_t)r * L + t];
    } else {
        const float* x = logits + (int64_t)r * ld;
        float best = -INFINITY;
        int idx = 0x7fffffff;
        for (int j = threadIdx.x; j < V; j += blockDim.x) {
            const float v = x[j];
            if (v > best || (v == best && j < idx)) { best = v; idx = j; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(idx, o, 64);
            if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
        }
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { bv[w] = best; bi[w] = idx; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int k = 1; k < 4; ++k)
                if (bv[k] > best || (bv[k] == best && bi[k] < idx)) { best = bv[k]; idx = bi[k]; }
            chosen = idx == 0x7fffffff ? 0 : idx;        // no maximum (a row of NaN / -inf): word 0, not an out-of-range row of E
        }
    }
    __syncthreads();
    const int64_t id = chosen;
    if (threadIdx.x == 0) ids_out[r] = id;
    for (int j = threadIdx.x; j < W; j += blockDim.x) {
        float v = E[id * W + j];
        if (p > 0.f) v *= drop_scale(seed, site, (uint64_t)(row0 + r) * W + j, p);
        out[(int64_t)r * ldo + j] = v;
    }
}

__global__ void copy2d_kernel(const float* __restrict__ src, int64_t lds_, float* __restrict__ dst, int64_t ldd, int rows,
                              int n, int accum) {
    const int64_t total = (int64_t)rows * n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / n;
        const int j = (int)(i - r * n);
        const float v = src[r * lds_ + j];
        float* d = dst + r * ldd + j;
        *d = accum ? *d + v : v;
    }
}
__global__ void dropout_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy, int rows, int n,
                               float p, uint64_t seed, uint32_t site, const uint64_t* seed_ptr) {
    if (seed_ptr) seed += *seed_ptr;
    const int64_t total = (int64_t)rows * n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / n;
        const int j = (int)(i - r * n);
        y[r * ldy + j] = x[r * ldx + j] * drop_scale(seed, site, (uint64_t)i, p);
    }
}
__global__ void permute_tb_kernel(const float* __restrict__ src, float* __restrict__ dst, int T, int B, int n) {
    const int64_t total = (int64_t)T * B * n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % n);
        const int64_t rb = i / n;           // destination row b*T + t
        const int t = (int)(rb % T), b = (int)(rb / T);
        dst[i] = src[((int64_t)t * B + b) * n + j];
    }
}
// dst[r, :] = src[idx[r], :].  Rows here are from a 2-KB embedding row to the 3.4-MB feature block of a clip (the HBM-resident
// feature store's batch gather), so a row is spread over blockIdx.x in 16-KB pieces: 64 rows alone would leave 3/4 of the CUs idle.
template <bool VEC>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, int64_t lds_, const int64_t* __restrict__ idx,
                                                          float* __restrict__ dst, int64_t ldd, int n) {
    const int pieces = (n + 4095) >> 12;
    const int r = blockIdx.x / pieces, piece = blockIdx.x - r * pieces;
    const float* s = src + idx[r] * lds_;
    float* d = dst + (int64_t)r * ldd;
    if constexpr (VEC) {
        const int n4 = n >> 2;
        const int base = piece * 1024 + threadIdx.x;
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (base + u * 256 < n4) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(s) + base + u * 256);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (base + u * 256 < n4) reinterpret_cast<f4*>(d)[base + u * 256] = v[u];
    } else {
        const int hi = min(n, (piece + 1) * 4096);
        for (int j = piece * 4096 + threadIdx.x; j < hi; j += 256) d[j] = s[j];
    }
}
__global__ void fill_kernel(float* __restrict__ dst, int64_t n, float v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = v;
}

// ------------------------------------------------------------------------------------------------ ragged CE
__global__ __launch_bounds__(256) void ce_ragged_kernel(const float* __restrict__ logits, const int64_t* __restrict__ targets,
                                                        const int64_t* __restrict__ lens, float* __restrict__ dlogits,
                                                        float* __restrict__ row_loss, int B, int L, int V, int tm) {
    __shared__ float red[16];
    const int row = blockIdx.x;
    const int b = tm ? row % B : row / L, t = tm ? row / B : row % L;
    const float* x = logits + (int64_t)row * V;
    float* dx = dlogits + (int64_t)row * V;
    int64_t len = lens[b];
    if (len > L) len = L;
    if (t >= len) {
        for (int j = threadIdx.x; j < V; j += blockDim.x) dx[j] = 0.f;
        if (threadIdx.x == 0) row_loss[row] = 0.f;
        return;
    }
    int64_t ntot = 0;
    for (int i = 0; i < B; ++i) ntot += (lens[i] > L ? L : lens[i]);
    float m = -INFINITY;
    for (int j = threadIdx.x; j < V; j += blockDim.x) m = fmaxf(m, x[j]);
    m = block_max(m, red);
    float s = 0.f;
    for (int j = threadIdx.x; j < V; j += blockDim.x) s += __expf(x[j] - m);
    s = block_sum(s, red);
    const int64_t tgt = targets[(int64_t)b * L + t];
    // a target outside [0, V) (torch's CrossEntropyLoss raises): no out-of-bounds read; the row's loss -- hence the step's
    // loss -- and its gradient are NaN, which no caller can miss
    const bool bad = tgt < 0 || tgt >= V;
    const float inv = bad ? NAN : 1.f / s, invn = 1.f / (float)ntot;
    for (int j = threadIdx.x; j < V; j += blockDim.x) {
        float p = __expf(x[j] - m) * inv;
        if (j == tgt) p -= 1.f;
        dx[j] = p * invn;
    }
    if (threadIdx.x == 0) row_loss[row] = bad ? NAN : (logf(s) + m - x[tgt]) * invn;
}
__global__ __launch_bounds__(256) void sum_to_scalar_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += v[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = s;
}

__global__ __launch_bounds__(256) void log_softmax_kernel(const float* __restrict__ logits, float* __restrict__ out, int V) {
    __shared__ float red[16];
    const float* x = logits + (int64_t)blockIdx.x * V;
    float* y = out + (int64_t)blockIdx.x * V;
    float m = -INFINITY;
    for (int j = threadIdx.x; j < V; j += blockDim.x) m = fmaxf(m, x[j]);
    m = block_max(m, red);
    float s = 0.f;
    for (int j = threadIdx.x; j < V; j += blockDim.x) s += expf(x[j] - m);
    s = block_sum(s, red);
    const float lse = logf(s) + m;
    for (int j = threadIdx.x; j < V; j += blockDim.x) y[j] = x[j] - lse;
}

// ------------------------------------------------------------------------------------------------ beam search step
// One workgroup per batch item, one wave per live beam (BeamSearch.search, allennlp_beamsearch.py:140-260 with
// per_node_beam_size == beam_size): log-softmax of the beam's logits row, its top-k classes (a finished beam offers
// <end> at log-prob 0 and nothing else), then the top-k of the k*k summed candidates, back-pointers and the row
// indices that reorder the recurrent state.  Ties resolve to the lower index.
constexpr int BEAM_MAXK = 8;

// (value, index) order of the search: larger value first, ties to the smaller index (torch.topk on equal log-probs is not
// specified; the oracle and the reference fixtures agree with this rule on every golden case)
__device__ __forceinline__ bool beam_better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
__device__ __forceinline__ int wave_min_i32(int v) {
    v = min(v, dpp_i32<0xB1>(v));
    v = min(v, dpp_i32<0x4E>(v));
    v = min(v, dpp_i32<0x141>(v));
    v = min(v, dpp_i32<0x140>(v));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
// the best (value, index) pair of the wave, to every lane: two DPP reductions instead of twelve ds_bpermute round trips
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
    const float m = dlsg::wave_max(v);
    i = wave_min_i32(v == m ? i : 0x7fffffff);
    v = m;
}

// One row of the search: the K best classes of a V-wide logit row and the row's log-sum-exp, ONE wave.  The row is read twice
// (maximum + each lane's own K best in the first pass, the exponential sum in the second -- same summation order as a plain
// strided loop, so the log-probs are bit-identical to the previous 2 + K pass form), then K merge rounds over the lanes' heads.
template <int K>
__device__ __forceinline__ void beam_row_topk(const float* x, int V, int lane, float& lse, float (&outv)[K], int (&outi)[K]) {
    float tv[K];
    int ti[K];
#pragma unroll
    for (int q = 0; q < K; ++q) { tv[q] = -INFINITY; ti[q] = 0x7fffffff; }
    float m = -INFINITY;
    for (int j = lane; j < V; j += 64) {
        float cv = x[j];
        int ci = j;
        m = fmaxf(m, cv);
        if (beam_better(cv, ci, tv[K - 1], ti[K - 1])) {
#pragma unroll
            for (int q = 0; q < K; ++q) {                       // one bubble pass keeps tv sorted
                const bool up = beam_better(cv, ci, tv[q], ti[q]);
                const float nv = up ? tv[q] : cv;
                const int ni = up ? ti[q] : ci;
                tv[q] = up ? cv : tv[q];
                ti[q] = up ? ci : ti[q];
                cv = nv; ci = ni;
            }
        }
    }
    m = dlsg::wave_max(m);
    float sum = 0.f;
    for (int j = lane; j < V; j += 64) sum += expf(x[j] - m);
    sum = dlsg::wave_sum(sum);
    lse = logf(sum) + m;
#pragma unroll
    for (int c = 0; c < K; ++c) {
        float bv = tv[0];
        int bi = ti[0];
        wave_argmax(bv, bi);
        outv[c] = bv; outi[c] = bi;
        if (ti[0] == bi && bi != 0x7fffffff) {                  // the winning lane pops its head
#pragma unroll
            for (int q = 0; q + 1 < K; ++q) { tv[q] = tv[q + 1]; ti[q] = ti[q + 1]; }
            tv[K - 1] = -INFINITY; ti[K - 1] = 0x7fffffff;
        }
    }
}

template <int K>
__global__ __launch_bounds__(64 * K) void beam_select_kernel(const dlsg_beam_select_args a) {
    __shared__ float cand_lp[K * K];
    __shared__ int cand_cls[K * K];
    const int b = blockIdx.x, k = K, V = a.V;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nbeam = a.first ? 1 : k;                      // step 0: the k rows of a group are identical, use row 0
    if (w < nbeam) {
        const int64_t row = (int64_t)b * k + w;
        const float* x = a.logits + row * a.ld;
        const bool ended = !a.first && a.last[row] == a.end;
        const float base = a.first ? 0.f : a.last_lp[row];
        if (ended) {
            if (lane < k) {
                cand_lp[w * k + lane] = lane == 0 ? base : -INFINITY;
                cand_cls[w * k + lane] = lane == 0 ? a.end : (lane - 1 < a.end ? lane - 1 : lane);
            }
        } else {
            float lse, bv[K];
            int bi[K];
            beam_row_topk<K>(x, V, lane, lse, bv, bi);
            if (lane == 0) {
#pragma unroll
                for (int c = 0; c < K; ++c) {
                    cand_lp[w * k + c] = (bv[c] - lse) + base;
                    cand_cls[w * k + c] = bi[c];
                }
            }
        }
    }
    __syncthreads();
    if (w == 0) {
        const int n = nbeam * k;
        float v = lane < n ? cand_lp[lane] : -INFINITY;
        int idx = lane < n ? lane : 0x7fffffff;
        int n_end = 0;
        for (int c = 0; c < k; ++c) {
            float bv = v;
            int bi = idx;
            wave_argmax(bv, bi);
            if (lane == bi) v = -INFINITY, idx = 0x7fffffff;        // remove the winner (it can win only once)
            if (bi == 0x7fffffff) bi = c < n ? c : 0;               // fewer than k finite candidates: any slot, log-prob -inf
            const int cls = cand_cls[bi];
            if (lane == 0) {
                const int64_t o = (int64_t)b * k + c;
                a.pred[o] = cls;
                a.new_lp[o] = bv;
                a.back[o] = bi / k;
                a.rows[o] = (int64_t)b * k + bi / k;
            }
            n_end += cls == a.end;
        }
        if (lane == 0 && a.ended_count) atomicAdd(a.ended_count, n_end);
    }
}

// dst_i[r, :] = src_i[rows[r], :] for up to four (src, dst, width) triples in one launch (blockIdx.y selects)
__global__ void gather_rows_multi_kernel(const dlsg_gather_multi_args a) {
    const int i = blockIdx.y, r = blockIdx.x;
    const int n = a.n[i];
    const float* s = a.src[i] + a.rows[r] * (int64_t)n;
    float* d = a.dst[i] + (int64_t)r * n;
    if ((n & 3) == 0) {
        for (int j = threadIdx.x * 4; j < n; j += blockDim.x * 4)
            *reinterpret_cast<f32x4*>(d + j) = *reinterpret_cast<const f32x4*>(s + j);
    } else {
        for (int j = threadIdx.x; j < n; j += blockDim.x) d[j] = s[j];
    }
}

// ------------------------------------------------------------------------------------------------ Adam
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float lr_bc1, float b1, float b2, float eps,
                                         float bc2s, float gscale) {
    const float gi = g * gscale;
    m = b1 * m + (1.f - b1) * gi;
    v = b2 * v + (1.f - b2) * gi * gi;
    const float denom = sqrtf(v) / bc2s + eps;
    p -= lr_bc1 * (m / denom);
}
// NTM: bit 0 non-temporal loads of m, v, g; bit 1 non-temporal stores of m, v; bit 2 / 3 the same for the load / store of p.
// 28 B of traffic per parameter: 16-byte accesses over the aligned body (the four arrays share one index, so one alignment),
// scalar head and tail (a trainable range may start at any parameter boundary)
template <int NTM>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                   float bc1, float bc2s, float gscale, const float* __restrict__ hyper,
                                                   const int32_t* __restrict__ guard) {
    if (guard && *guard != 0) return;      // a persistent kernel reported a time-out in this step: its gradients are invalid
    if (hyper) { lr = hyper[0]; bc1 = 1.f; bc2s = hyper[1]; }
    const float lr_bc1 = lr / bc1;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
    const bool same = ((reinterpret_cast<uintptr_t>(p) ^ reinterpret_cast<uintptr_t>(g)) & 15) == 0 &&
                      ((reinterpret_cast<uintptr_t>(p) ^ reinterpret_cast<uintptr_t>(m)) & 15) == 0 &&
                      ((reinterpret_cast<uintptr_t>(p) ^ reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    int64_t head = same ? (int64_t)((16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15) / 4 : n;
    if (head > n) head = n;
    const int64_t nv = (n - head) / 4;                     // float4 elements of the aligned body
    for (int64_t i = tid; i < head; i += nth) adam_one(p[i], g[i], m[i], v[i], lr_bc1, b1, b2, eps, bc2s, gscale);
    f32x4* p4 = reinterpret_cast<f32x4*>(p + head);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g + head);
    f32x4* m4 = reinterpret_cast<f32x4*>(m + head);
    f32x4* v4 = reinterpret_cast<f32x4*>(v + head);
    for (int64_t i = tid; i < nv; i += nth) {
        f32x4 pp, mm, vv, gg;
        if (NTM & 4) pp = __builtin_nontemporal_load(p4 + i); else pp = p4[i];
        if (NTM & 1) { mm = __builtin_nontemporal_load(m4 + i); vv = __builtin_nontemporal_load(v4 + i); gg = __builtin_nontemporal_load(g4 + i); }
        else { mm = m4[i]; vv = v4[i]; gg = g4[i]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float pe = pp[e], me = mm[e], ve = vv[e];
            adam_one(pe, gg[e], me, ve, lr_bc1, b1, b2, eps, bc2s, gscale);
            pp[e] = pe; mm[e] = me; vv[e] = ve;
        }
        if (NTM & 2) { __builtin_nontemporal_store(mm, m4 + i); __builtin_nontemporal_store(vv, v4 + i); }
        else { m4[i] = mm; v4[i] = vv; }
        if (NTM & 8) __builtin_nontemporal_store(pp, p4 + i); else p4[i] = pp;
    }
    for (int64_t i = head + 4 * nv + tid; i < n; i += nth) adam_one(p[i], g[i], m[i], v[i], lr_bc1, b1, b2, eps, bc2s, gscale);
}

inline int grid_for(int64_t total, int threads = 256, int cap = 4096) {
    int64_t b = (total + threads - 1) / threads;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}
#define ST(s) reinterpret_cast<hipStream_t>(s)

}  // namespace

extern "C" int dlsg_rowln_fwd_multi(const dlsg_rowln_args* a, int count, void* stream) {
    if (!a || count < 1 || count > DLSG_ROWLN_MAXMULTI) return DLSG_EINVAL;
    RowLnPack pk;
    for (int i = 0; i < count; ++i) {
        if (a[i].n < 1 || a[i].n > LN_THREADS * LN_MAXPT || a[i].rows != a[0].rows) return DLSG_EINVAL;
        pk.s[i] = a[i];
    }
    if (a->rows == 0) return DLSG_OK;
    int grid = a->rows < 4096 ? a->rows : 4096;
    hipLaunchKernelGGL(rowln_fwd_kernel, dim3(grid, count), dim3(LN_THREADS), 0, ST(stream), pk);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_rowln_fwd(const dlsg_rowln_args* a, void* stream) { return dlsg_rowln_fwd_multi(a, 1, stream); }
// grid of the backward = number of per-block dgamma/dbeta partial rows: enough blocks to fill the chip on tall inputs
extern "C" int dlsg_rowln_bwd_nblk(int rows) {
    if (rows < 1) return 1;
    if (rows <= 256) return rows;
    if (rows <= 4096) return (rows + 1) / 2 < 1024 ? (rows + 1) / 2 : 1024;   // two rows per block: 1664-row norms ran at 0.6 TB/s on 256
    return 1024;
}
extern "C" int dlsg_rowln_bwd_multi(const dlsg_rowln_bwd_args* a, int count, void* stream) {
    if (!a || count < 1 || count > DLSG_ROWLN_MAXMULTI) return DLSG_EINVAL;
    const int grid = dlsg_rowln_bwd_nblk(a->f.rows);
    RowLnBwdPack pk;
    for (int i = 0; i < count; ++i) {
        if (a[i].f.n < 1 || a[i].f.n > LN_THREADS * LN_MAXPT || a[i].f.rows != a[0].f.rows) return DLSG_EINVAL;
        if (a[i].dgb_part && a[i].nblk != grid) return DLSG_EINVAL;
        pk.s[i] = a[i];
    }
    if (a->f.rows == 0) return DLSG_OK;
    hipLaunchKernelGGL(rowln_bwd_kernel, dim3(grid, count), dim3(LN_THREADS), 0, ST(stream), pk);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_rowln_bwd(const dlsg_rowln_bwd_args* a, void* stream) { return dlsg_rowln_bwd_multi(a, 1, stream); }
static int colsum_chunks(int rows) {
    if (rows < 4096) return 1;
    const int chunks = (rows + 1023) / 1024;
    return chunks > 32 ? 32 : chunks;
}
static int colsum_launch(const float* part, int64_t ld, int rows, int n, float* out, float* out2, int split, int mode, int accum,
                         float* ws, void* stream) {
    if (!part || !out || n < 1) return DLSG_EINVAL;
    const int chunks = colsum_chunks(rows);
    const int rpc = (rows + chunks - 1) / chunks;
    const bool v4 = n % 4 == 0 && ld % 4 == 0 && (reinterpret_cast<uintptr_t>(part) & 15) == 0;
    if (chunks > 1 && ws && (v4 || !out2)) {
        // tall input with a workspace: chunk partials, then a fixed-order combine
        if (v4) hipLaunchKernelGGL(colsum_v4_kernel, dim3((n + 63) / 64, chunks), dim3(1024), 0, ST(stream), part, ld, rows, n, out,
                                   out2, out2 ? split : n, out2 ? mode : 0, accum, rpc, ws);
        else hipLaunchKernelGGL(colsum_kernel, dim3((n + 63) / 64, chunks), dim3(1024), 0, ST(stream), part, ld, rows, n, out, accum,
                                rpc, ws);
        hipLaunchKernelGGL(colsum_finish_kernel, dim3((n + 255) / 256), dim3(256), 0, ST(stream), ws, chunks, n, out, out2,
                           out2 ? split : n, out2 ? mode : 0, accum);
        DLSG_CHECK_LAUNCH();
        return DLSG_OK;
    }
    if (chunks > 1 && !accum) {
        const int n0 = (out2 && mode == 0) ? split : n;
        hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(256), 0, ST(stream), out, (int64_t)n0, 0.f);
        if (out2) hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(256), 0, ST(stream), out2, (int64_t)(mode == 0 ? n - split : n), 0.f);
    }
    float* none = nullptr;
    if (v4) {
        hipLaunchKernelGGL(colsum_v4_kernel, dim3((n + 63) / 64, chunks), dim3(1024), 0, ST(stream), part, ld, rows, n, out,
                           out2, out2 ? split : n, out2 ? mode : 0, accum, rpc, none);
    } else if (!out2) {
        hipLaunchKernelGGL(colsum_kernel, dim3((n + 63) / 64, chunks), dim3(1024), 0, ST(stream), part, ld, rows, n, out, accum, rpc, none);
    } else if (mode == 0) {      // scalar fallback: the two halves / destinations as separate launches
        hipLaunchKernelGGL(colsum_kernel, dim3((split + 63) / 64, chunks), dim3(1024), 0, ST(stream), part, ld, rows, split, out,
                           accum, rpc, none);
        hipLaunchKernelGGL(colsum_kernel, dim3((n - split + 63) / 64, chunks), dim3(1024), 0, ST(stream), part + split, ld, rows,
                           n - split, out2, accum, rpc, none);
    } else {
        hipLaunchKernelGGL(colsum_kernel, dim3((n + 63) / 64, chunks), dim3(1024), 0, ST(stream), part, ld, rows, n, out, accum, rpc, none);
        hipLaunchKernelGGL(colsum_kernel, dim3((n + 63) / 64, chunks), dim3(1024), 0, ST(stream), part, ld, rows, n, out2, accum, rpc, none);
    }
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_colsum_multi_ok(const float* part, int64_t ld, int rows, int n) {
    return (colsum_chunks(rows) == 1 && n % 4 == 0 && ld % 4 == 0 && (reinterpret_cast<uintptr_t>(part) & 15) == 0) ? 1 : 0;
}
extern "C" int dlsg_colsum_multi(const dlsg_colsum_desc* d, int count, void* stream) {
    if (!d || count < 1 || count > DLSG_COLSUM_MAXMULTI) return DLSG_EINVAL;
    ColsumPack pk;
    int maxn = 0;
    for (int i = 0; i < count; ++i) {
        if (!d[i].part || !d[i].out_a || d[i].n < 1 || d[i].rows < 0) return DLSG_EINVAL;
        if (!dlsg_colsum_multi_ok(d[i].part, d[i].ld, d[i].rows, d[i].n)) return DLSG_EALIGN;
        if (d[i].out_b && !d[i].dup && (d[i].split < 0 || d[i].split > d[i].n)) return DLSG_EINVAL;
        pk.d[i] = d[i];
        maxn = d[i].n > maxn ? d[i].n : maxn;
    }
    hipLaunchKernelGGL(colsum_multi_kernel, dim3((maxn + 63) / 64, count), dim3(1024), 0, ST(stream), pk);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int64_t dlsg_colsum_ws_floats(int rows, int n) {
    const int chunks = colsum_chunks(rows);
    return chunks > 1 ? (int64_t)chunks * n : 0;
}
extern "C" int dlsg_colsum(const float* part, int64_t ld, int rows, int n, float* out, int accum, float* ws, void* stream) {
    return colsum_launch(part, ld, rows, n, out, nullptr, n, 0, accum, ws, stream);
}
extern "C" int dlsg_colsum2(const float* part, int64_t ld, int rows, int n, float* out_a, float* out_b, int split, int dup,
                            int accum, float* ws, void* stream) {
    if (!out_b || (!dup && (split < 0 || split > n))) return DLSG_EINVAL;
    return colsum_launch(part, ld, rows, n, out_a, out_b, dup ? n : split, dup ? 1 : 0, accum, ws, stream);
}
extern "C" int dlsg_softmax_fwd(const float* x, const float* mask, float* y, int64_t outer, int n, int inner, void* stream) {
    const int64_t lines = outer * inner;
    if (lines == 0) return DLSG_OK;
    hipLaunchKernelGGL(softmax_fwd_kernel, dim3((unsigned)((lines + 3) / 4)), dim3(256), 0, ST(stream), x, mask, y, outer, n,
                       inner);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_softmax_bwd(const float* y, const float* dy, float* dx, int64_t outer, int n, int inner, void* stream) {
    const int64_t lines = outer * inner;
    if (lines == 0) return DLSG_OK;
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)((lines + 3) / 4)), dim3(256), 0, ST(stream), y, dy, dx, outer, n,
                       inner);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_lstm_pw_fwd_n(const dlsg_lstm_pw_args* a, int count, void* stream) {
    if (!a || count < 1 || count > 2 || a[0].H < 1) return DLSG_EINVAL;
    if (count == 2 && (a[1].H != a[0].H || a[1].B != a[0].B)) return DLSG_EINVAL;
    if (a[0].B == 0) return DLSG_OK;
    PwFwdSet set;
    for (int i = 0; i < 2; ++i) set.a[i] = a[i < count ? i : 0];
    hipLaunchKernelGGL(lstm_pw_fwd_kernel, dim3((a[0].H + 255) / 256, a[0].B, count), dim3(256), 0, ST(stream), set);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_lstm_pw_fwd(const dlsg_lstm_pw_args* a, void* stream) { return dlsg_lstm_pw_fwd_n(a, 1, stream); }
extern "C" int dlsg_lstm_pw_bwd_n(const dlsg_lstm_pw_bwd_args* a, int count, void* stream) {
    if (!a || count < 1 || count > 2 || a[0].H < 1) return DLSG_EINVAL;
    if (count == 2 && (a[1].H != a[0].H || a[1].B != a[0].B)) return DLSG_EINVAL;
    if (a[0].B == 0) return DLSG_OK;
    PwBwdSet set;
    for (int i = 0; i < 2; ++i) set.a[i] = a[i < count ? i : 0];
    hipLaunchKernelGGL(lstm_pw_bwd_kernel, dim3((a[0].H + 255) / 256, a[0].B, count), dim3(256), 0, ST(stream), set);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_lstm_pw_bwd(const dlsg_lstm_pw_bwd_args* a, void* stream) { return dlsg_lstm_pw_bwd_n(a, 1, stream); }
extern "C" int dlsg_mean_rows_fwd(const float* x, float* out, int64_t ldo, int B, int P, int H, void* stream) {
    if (B == 0) return DLSG_OK;
    hipLaunchKernelGGL(mean_rows_fwd_kernel, dim3((H + 255) / 256, B), dim3(256), 0, ST(stream), x, out, ldo, P, H);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_mean_rows_bwd(const float* dout, int64_t lddo, float* dx, int B, int P, int H, int accum, void* stream) {
    if (B == 0) return DLSG_OK;
    hipLaunchKernelGGL(mean_rows_bwd_kernel, dim3((H + 255) / 256, B), dim3(256), 0, ST(stream), dout, lddo, dx, P, H, accum);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_embed_fwd(const float* E, const int64_t* ids, float* out, int64_t ldo, int rows, int W, float p,
                              uint64_t seed, uint32_t site, int64_t row0, const uint64_t* seed_ptr, void* stream) {
    if (rows == 0) return DLSG_OK;
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(rows), dim3(128), 0, ST(stream), E, ids, out, ldo, W, p, seed, site, row0, seed_ptr);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_embed_bwd(const float* dout, int64_t lddo, const int64_t* ids, float* dE, int rows, int W, float p,
                              uint64_t seed, uint32_t site, int64_t row0, const uint64_t* seed_ptr, void* stream) {
    if (rows == 0) return DLSG_OK;
    const int threads = W >= 1024 ? 1024 : (W + 63) / 64 * 64;           // one wave per 64 columns
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(rows), dim3(threads), 0, ST(stream), dout, lddo, ids, dE, rows, W, p, seed, site, row0,
                       seed_ptr);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_select_embed(const float* logits, int64_t ld, int V, const int64_t* captions, int L, int t,
                                 const int32_t* coins, const float* E, int64_t* ids_out, float* out, int64_t ldo, int rows, int W,
                                 float p, uint64_t seed, uint32_t site, int64_t row0, const uint64_t* seed_ptr, int prefilled,
                                 void* stream) {
    if (rows == 0) return DLSG_OK;
    hipLaunchKernelGGL(select_embed_kernel, dim3(rows), dim3(256), 0, ST(stream), logits, ld, V, captions, L, t, coins, E, ids_out,
                       out, ldo, W, p, seed, site, row0, seed_ptr, prefilled);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_argmax(const float* logits, int64_t ld, int64_t* ids, int rows, int V, void* stream) {
    if (rows == 0) return DLSG_OK;
    hipLaunchKernelGGL(argmax_kernel, dim3(rows), dim3(256), 0, ST(stream), logits, ld, ids, V);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_copy2d(const float* src, int64_t lds_, float* dst, int64_t ldd, int rows, int n, int accum, void* stream) {
    if ((int64_t)rows * n == 0) return DLSG_OK;
    hipLaunchKernelGGL(copy2d_kernel, dim3(grid_for((int64_t)rows * n)), dim3(256), 0, ST(stream), src, lds_, dst, ldd, rows, n,
                       accum);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_dropout(const float* x, int64_t ldx, float* y, int64_t ldy, int rows, int n, float p, uint64_t seed,
                            uint32_t site, const uint64_t* seed_ptr, void* stream) {
    if ((int64_t)rows * n == 0) return DLSG_OK;
    hipLaunchKernelGGL(dropout_kernel, dim3(grid_for((int64_t)rows * n)), dim3(256), 0, ST(stream), x, ldx, y, ldy, rows, n, p,
                       seed, site, seed_ptr);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_gather_rows(const float* src, int64_t lds_, const int64_t* idx, float* dst, int64_t ldd, int rows, int n,
                                void* stream) {
    if (rows == 0 || n == 0) return DLSG_OK;
    const bool vec = n % 4 == 0 && lds_ % 4 == 0 && ldd % 4 == 0 && ((uintptr_t)src | (uintptr_t)dst) % 16 == 0;
    const dim3 grid((unsigned)((int64_t)((n + 4095) / 4096) * rows));
    if (vec) hipLaunchKernelGGL(gather_rows_kernel<true>, grid, dim3(256), 0, ST(stream), src, lds_, idx, dst, ldd, n);
    else hipLaunchKernelGGL(gather_rows_kernel<false>, grid, dim3(256), 0, ST(stream), src, lds_, idx, dst, ldd, n);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_fill(float* dst, int64_t n, float value, void* stream) {
    if (n == 0) return DLSG_OK;
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, ST(stream), dst, n, value);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_permute_tb(const float* src, float* dst, int T, int B, int n, void* stream) {
    const int64_t total = (int64_t)T * B * n;
    if (total == 0) return DLSG_OK;
    hipLaunchKernelGGL(permute_tb_kernel, dim3(grid_for(total)), dim3(256), 0, ST(stream), src, dst, T, B, n);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_ce_ragged(const float* logits, const int64_t* targets, const int64_t* lens, float* dlogits, float* row_loss,
                              float* loss, int B, int L, int V, int time_major, void* stream) {
    if (B * L == 0) return DLSG_OK;
    hipLaunchKernelGGL(ce_ragged_kernel, dim3(B * L), dim3(256), 0, ST(stream), logits, targets, lens, dlogits, row_loss, B, L, V,
                       time_major);
    hipLaunchKernelGGL(sum_to_scalar_kernel, dim3(1), dim3(256), 0, ST(stream), row_loss, B * L, loss);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_log_softmax(const float* logits, float* out, int rows, int V, void* stream) {
    if (rows == 0) return DLSG_OK;
    hipLaunchKernelGGL(log_softmax_kernel, dim3(rows), dim3(256), 0, ST(stream), logits, out, V);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_beam_select(const dlsg_beam_select_args* a, void* stream) {
    if (!a || a->k < 1 || a->k > BEAM_MAXK || a->V < a->k) return DLSG_EINVAL;
    if (a->B == 0) return DLSG_OK;
    switch (a->k) {
        case 1: hipLaunchKernelGGL(beam_select_kernel<1>, dim3(a->B), dim3(64), 0, ST(stream), *a); break;
        case 2: hipLaunchKernelGGL(beam_select_kernel<2>, dim3(a->B), dim3(128), 0, ST(stream), *a); break;
        case 3: hipLaunchKernelGGL(beam_select_kernel<3>, dim3(a->B), dim3(192), 0, ST(stream), *a); break;
        case 4: hipLaunchKernelGGL(beam_select_kernel<4>, dim3(a->B), dim3(256), 0, ST(stream), *a); break;
        case 5: hipLaunchKernelGGL(beam_select_kernel<5>, dim3(a->B), dim3(320), 0, ST(stream), *a); break;
        case 6: hipLaunchKernelGGL(beam_select_kernel<6>, dim3(a->B), dim3(384), 0, ST(stream), *a); break;
        case 7: hipLaunchKernelGGL(beam_select_kernel<7>, dim3(a->B), dim3(448), 0, ST(stream), *a); break;
        default: hipLaunchKernelGGL(beam_select_kernel<8>, dim3(a->B), dim3(512), 0, ST(stream), *a); break;
    }
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_gather_rows_multi(const dlsg_gather_multi_args* a, void* stream) {
    if (!a || a->count < 1 || a->count > 4) return DLSG_EINVAL;
    if (a->nrows == 0) return DLSG_OK;
    hipLaunchKernelGGL(gather_rows_multi_kernel, dim3(a->nrows, a->count), dim3(256), 0, ST(stream), *a);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                         int step, float grad_scale, const float* hyper, const int32_t* guard, void* stream) {
    if (n == 0) return DLSG_OK;
    const float bc1 = 1.f - powf(b1, (float)step);
    const float bc2s = sqrtf(1.f - powf(b2, (float)step));
    // moments and gradients are read once per step and the moments written once: non-temporal on those streams (mask 3), default
    // policy on the parameters, which the next forward reads.  Measured per launch under rocprofv3 on two boxes: 502 -> 424 us and
    // 421 -> 391 us; every other combination of the four streams within 391-426 us on the second box
    hipLaunchKernelGGL(adam_kernel<3>, dim3(grid_for((n + 3) / 4, 256, 8192)), dim3(256), 0, ST(stream), p, g, m, v, n, lr, b1, b2, eps, bc1,
                       bc2s, grad_scale, hyper, guard);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
