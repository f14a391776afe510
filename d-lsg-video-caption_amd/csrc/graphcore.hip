// The two small per-clip graphs of the encoder, each as ONE launch with one workgroup per clip (a4, a6):
//   * latent_psl_fwd : LatentPSL (reference models/sublayer.py:189-198): logits = ov . theta^T, softmax over the frames,
//                      u = adj^T ov, Dropout(LayerNorm(tanh(u))).  Unfused it is two batched GEMMs whose output tiles are
//                      26 x 8 (a 64 x 64 MFMA tile is 95 % padding), a softmax and a LayerNorm launch.
//   * sa_core_fwd    : the 26 x 26 core of SelfAttention (sublayer.py:69-78): logits = K Q^T * scale, softmax over the
//                      Q index, out = w V.  Unfused: batched GEMM + softmax + batched GEMM.
// Both are HBM-bound in isolation (0.14 MB and 0.85 MB per clip); the unfused launches ran them at 0.3 / 0.6 TB/s.
#include <mutex>

#include "common.hpp"
#include "dlsg.h"

using namespace dlsg;

namespace {

constexpr int PSL_THREADS = 1024;
constexpr int PSL_MAXT = 32, PSL_MAXP = 32;

__device__ __forceinline__ int crow(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

// ------------------------------------------------------------------------------------------------ LatentPSL forward
// NX = float4 per lane of one frame row (H <= 256 NX); NX = 4 (H <= 1024) fits 64 registers: two 1024-thread workgroups per CU
// (both encoder streams in one launch: blockIdx.y picks the argument block -- 64 clips put 64 workgroups on 256 CUs)
struct PslPack { dlsg_latent_psl_args s[DLSG_PSL_MAXMULTI]; };
struct PslBwdPack { dlsg_latent_psl_bwd_args s[DLSG_PSL_MAXMULTI]; };

template <int NX>
__global__ __launch_bounds__(PSL_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void latent_psl_fwd_kernel(const PslPack pk) {
    const dlsg_latent_psl_args& a = pk.s[blockIdx.y];
    extern __shared__ __attribute__((aligned(16))) float smem[];        // theta: [P][H]
    __shared__ float lgs[PSL_MAXT][PSL_MAXP + 1];       // logits, then adj
    __shared__ float red[16 * 8];
    const int b = blockIdx.x;
    const int T = a.T, P = a.P, H = a.H;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t seed = a.seed + (a.seed_ptr ? *a.seed_ptr : 0ull);
    // The clip's frame nodes are NOT staged in LDS (106 KB at T = 26, H = 1024 allowed one workgroup per CU, whose load,
    // logits, aggregation and LayerNorm phases then ran with nothing to overlap them: 1.0 TB/s).  The logits pass reads every
    // frame row once from HBM straight into registers; the aggregation pass reads the rows again, from L2 (one clip is 106 KB),
    // with lanes along the columns.  theta (P x H) is what sits in LDS, so two workgroups share a CU.
    const float* ovl = a.ov + (int64_t)b * T * H;
    float* thl = smem;
    for (int i = threadIdx.x * 4; i < P * H; i += PSL_THREADS * 4)
        *reinterpret_cast<f32x4*>(thl + i) = *reinterpret_cast<const f32x4*>(a.theta + i);
    __syncthreads();
    // ---- logits[t][p] = ov[t] . theta[p]: a wave holds one frame row in registers and walks the proposals
    for (int t = w; t < T; t += PSL_THREADS / 64) {
        f32x4 x[NX];
#pragma unroll
        for (int c = 0; c < NX; ++c) {
            const int j = c * 256 + lane * 4;
            x[c] = (j < H) ? *reinterpret_cast<const f32x4*>(ovl + (int64_t)t * H + j) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        for (int p = 0; p < P; ++p) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < NX; ++c) {
                const int j = c * 256 + lane * 4;
                if (j < H) {
                    const f32x4 y = *reinterpret_cast<const f32x4*>(thl + p * H + j);
                    acc += x[c][0] * y[0] + x[c][1] * y[1] + x[c][2] * y[2] + x[c][3] * y[3];
                }
            }
            acc = wave_sum(acc);
            if (lane == 0) lgs[t][p] = acc;
        }
    }
    __syncthreads();
    // ---- softmax over the frames, per proposal
    if (threadIdx.x < P) {
        const int p = threadIdx.x;
        float m = -INFINITY;
        for (int t = 0; t < T; ++t) m = fmaxf(m, lgs[t][p]);
        float l = 0.f;
        for (int t = 0; t < T; ++t) l += __expf(lgs[t][p] - m);
        const float inv = 1.f / l;
        for (int t = 0; t < T; ++t) {
            const float v = __expf(lgs[t][p] - m) * inv;
            lgs[t][p] = v;
            a.adj[((int64_t)b * T + t) * P + p] = v;
        }
    }
    __syncthreads();
    // ---- u[p][h] = sum_t adj[t][p] ov[t][h];  psl = Dropout(LayerNorm(tanh(u))), proposals in chunks of 8
    for (int p0 = 0; p0 < P; p0 += 8) {
        float y[8][2];                                   // up to two columns per thread (H <= 2048)
        float s8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) s8[i] = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int h = threadIdx.x + c * PSL_THREADS;
            float acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = 0.f;
            if (h < H) {
                for (int t = 0; t < T; ++t) {
                    const float x = ovl[t * H + h];
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] += ((p0 + i < P) ? lgs[t][p0 + i] : 0.f) * x;
                }
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (p0 + i < P) a.u[((int64_t)b * P + p0 + i) * H + h] = acc[i];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) { y[i][c] = (h < H) ? tanhf(acc[i]) : 0.f; s8[i] += y[i][c]; }
        }
        // LayerNorm statistics of the 8 rows at once
#pragma unroll
        for (int i = 0; i < 8; ++i) s8[i] = wave_sum(s8[i]);
        __syncthreads();
        if (lane == 0)
#pragma unroll
            for (int i = 0; i < 8; ++i) red[16 * i + w] = s8[i];
        __syncthreads();
        float mean[8], q8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float r = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) r += red[16 * i + k];
            mean[i] = r / H;
            q8[i] = 0.f;
#pragma unroll
            for (int c = 0; c < 2; ++c)
                if (threadIdx.x + c * PSL_THREADS < H) { const float d = y[i][c] - mean[i]; q8[i] += d * d; }
            q8[i] = wave_sum(q8[i]);
        }
        __syncthreads();
        if (lane == 0)
#pragma unroll
            for (int i = 0; i < 8; ++i) red[16 * i + w] = q8[i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (p0 + i >= P) continue;
            float r = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) r += red[16 * i + k];
            const float rstd = rsqrtf(r / H + a.eps);
            const int64_t row = (int64_t)b * P + p0 + i;
            if (threadIdx.x == 0) { a.stats[2 * row] = mean[i]; a.stats[2 * row + 1] = rstd; }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int h = threadIdx.x + c * PSL_THREADS;
                if (h < H) {
                    float v = (y[i][c] - mean[i]) * rstd * a.gamma[h] + a.beta[h];
                    if (a.p > 0.f) v *= drop_scale(seed, a.site, (uint64_t)row * H + h, a.p);
                    a.out[row * H + h] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ self-attention core
// Workgroup = 8 waves, one clip.  Scores: K and Q are walked in 512-column chunks staged in LDS; wave w contracts its
// 64-column slice on the f32 matrix cores (A = K rows, B = Q rows), the 8 partial 32 x 32 tiles are summed through LDS
// (as in the o2v kernel).  Softmax along the Q index = across lanes of the C layout.  out = w V on the VALU: one thread
// per 4 columns, V rows read once with 16-B loads, w broadcast from LDS.
constexpr int SA_THREADS = 512;
constexpr int SA_CH = 512;
constexpr int SA_LD = SA_CH + 4;
constexpr int SA_LDS_FLOATS = 2 * 32 * SA_LD + 4 * 16 * 64;
// workgroups per clip for the column passes: up to D / 512, as many as keep the launch near one workgroup per CU
inline int sa_col_split(int B, int D) {
    int sp = 256 / (B > 0 ? B : 1);
    const int mx = D / SA_THREADS;
    if (sp > mx) sp = mx;
    return sp < 2 ? 1 : sp;
}

// rows [0, T) x columns [cbase, cbase + SA_CH) of a (T, D) array -> dst [32][SA_LD] (rows >= T and columns >= D zero)
__device__ __forceinline__ void sa_stage_rows(float* __restrict__ dst, const float* __restrict__ src, int T, int D, int cbase) {
    for (int f = threadIdx.x; f < 32 * (SA_CH / 4); f += SA_THREADS) {
        const int row = f / (SA_CH / 4), c4 = f % (SA_CH / 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < T && cbase + 4 * c4 < D) v = *reinterpret_cast<const f32x4*>(src + (int64_t)row * D + cbase + 4 * c4);
        *reinterpret_cast<f32x4*>(dst + row * SA_LD + 4 * c4) = v;
    }
}

__global__ __launch_bounds__(SA_THREADS) void sa_core_fwd_kernel(const dlsg_sa_core_args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float wl[32][33];
    float* kl = smem;                       // [32][SA_LD]
    float* ql = smem + 32 * SA_LD;
    float* red = smem + 2 * 32 * SA_LD;     // [4][16][64]
    const int b = blockIdx.x;
    const int T = a.T, D = a.D;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const float* Kb = a.K + (int64_t)b * T * D;
    const float* Qb = a.Q + (int64_t)b * T * D;
    const float* Vb = a.V + (int64_t)b * T * D;

    f32x16 sacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
    for (int c0 = 0; c0 < D; c0 += SA_CH) {
        const int cw = min(SA_CH, D - c0);              // multiple of 64
        __syncthreads();
        for (int f = threadIdx.x; f < 32 * (SA_CH / 4); f += SA_THREADS) {
            const int row = f / (SA_CH / 4), c4 = f % (SA_CH / 4);
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, qv = {0.f, 0.f, 0.f, 0.f};
            if (row < T && 4 * c4 < cw) {
                kv = *reinterpret_cast<const f32x4*>(Kb + (int64_t)row * D + c0 + 4 * c4);
                qv = *reinterpret_cast<const f32x4*>(Qb + (int64_t)row * D + c0 + 4 * c4);
            }
            *reinterpret_cast<f32x4*>(kl + row * SA_LD + 4 * c4) = kv;
            *reinterpret_cast<f32x4*>(ql + row * SA_LD + 4 * c4) = qv;
        }
        __syncthreads();
        // wave w: columns [64w, 64w+64) of the chunk; lane half h owns 32 of them
        if (64 * w < cw) {
            const float* ap = kl + r * SA_LD + 64 * w + 32 * h;
            const float* bp = ql + r * SA_LD + 64 * w + 32 * h;
#pragma unroll
            for (int s4 = 0; s4 < 8; ++s4) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap + 4 * s4);
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(bp + 4 * s4);
#pragma unroll
                for (int i = 0; i < 4; ++i) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i], b4[i], sacc, 0, 0, 0);
            }
        }
    }
    // ---- sum the 8 partial tiles (C layout: lane column = Q index j, registers = K index i)
    if (w >= 4) {
#pragma unroll
        for (int e = 0; e < 16; ++e) red[((w - 4) * 16 + e) * 64 + lane] = sacc[e];
    }
    __syncthreads();
    if (w < 4) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float t = sacc[e] + red[(w * 16 + e) * 64 + lane];
            red[(w * 16 + e) * 64 + lane] = t;
        }
    }
    __syncthreads();
    // ---- softmax over j (lanes of one half); wave w finishes registers 2w, 2w+1
#pragma unroll
    for (int ee = 0; ee < 2; ++ee) {
        const int e = 2 * w + ee;
        float sv = red[(0 * 16 + e) * 64 + lane] + red[(1 * 16 + e) * 64 + lane] + red[(2 * 16 + e) * 64 + lane] +
                   red[(3 * 16 + e) * 64 + lane];
        sv *= a.scale;
        const int i = crow(e, h);
        if (a.mask && i < T && r < T && !(a.mask[((int64_t)b * T + i) * T + r] > 0.f)) sv = -9e15f;
        if (r >= T) sv = -INFINITY;
        float m = sv;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        const float ex = __expf(sv - m);
        float l = ex;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) l += __shfl_xor(l, o, 64);
        const float wv = ex / l;
        wl[i][r] = wv;
        if (i < T && r < T && blockIdx.y == 0) a.w[((int64_t)b * T + i) * T + r] = wv;
    }
    __syncthreads();
    // ---- out[i][cols] = sum_j w[i][j] V[j][cols]
    if (gridDim.y > 1) {
        // few clips (B workgroups would leave 3/4 of the CUs idle): the clip's columns are spread over gridDim.y workgroups,
        // each of which has computed the (cheap) 26 x 26 weights itself; one column per thread
        // the 512 columns of V are staged through LDS first (16 B x 16 loads per thread in flight): read row by row from
        // global memory inside the product loop, each of the 26 rows cost its own round trip
        for (int cbase = blockIdx.y * SA_CH; cbase < D; cbase += gridDim.y * SA_CH) {
            __syncthreads();
            sa_stage_rows(kl, Vb, T, D, cbase);
            __syncthreads();
            const int c = cbase + threadIdx.x;
            if (c < D) {
                float acc[32];
#pragma unroll
                for (int i = 0; i < 32; ++i) acc[i] = 0.f;
                for (int j = 0; j < T; ++j) {
                    const float v = kl[j * SA_LD + threadIdx.x];
#pragma unroll
                    for (int i = 0; i < 32; ++i) acc[i] += wl[i][j] * v;
                }
#pragma unroll
                for (int i = 0; i < 32; ++i)
                    if (i < T) a.out[((int64_t)b * T + i) * D + c] = acc[i];
            }
        }
        return;
    }
    for (int c = threadIdx.x * 4; c < D; c += SA_THREADS * 4) {
        f32x4 acc[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < T; ++j) {
            const f32x4 v4 = *reinterpret_cast<const f32x4*>(Vb + (int64_t)j * D + c);
#pragma unroll
            for (int i = 0; i < 32; ++i) acc[i] += wl[i][j] * v4;
        }
#pragma unroll
        for (int i = 0; i < 32; ++i)
            if (i < T) *reinterpret_cast<f32x4*>(a.out + ((int64_t)b * T + i) * D + c) = acc[i];
    }
}

// ------------------------------------------------------------------------------------------------ self-attention core backward
// Given d(out) (B,T,D) and the saved softmax weights w (B,T,T): dw = d(out) V^T on the matrix cores (same chunked
// product as the forward's scores), softmax backward across the lanes of the C layout, then three column-parallel
// VALU passes   dV = w^T d(out),   dK = scale * dlg Q,   dQ = scale * dlg^T K.
// Replaces two batched GEMMs + softmax_bwd + two more batched GEMMs whose 26 x 26 x D products each launched
// thousands of nearly empty 64 x 64 tiles.
__global__ __launch_bounds__(SA_THREADS) void sa_core_bwd_kernel(const dlsg_sa_core_bwd_args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float wl[32][33];            // w[i][j]
    __shared__ float gl[32][33];            // scale * dlogits[i][j]
    float* al = smem;                       // [32][SA_LD]  d(out) rows
    float* vl = smem + 32 * SA_LD;          //              V rows
    float* red = smem + 2 * 32 * SA_LD;     // [4][16][64]
    const int b = blockIdx.x;
    const int T = a.T, D = a.D;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t base = (int64_t)b * T * D;
    const float* Gb = a.dout + base;
    const float* Kb = a.K + base;
    const float* Qb = a.Q + base;
    const float* Vb = a.V + base;

    f32x16 sacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
    for (int c0 = 0; c0 < D; c0 += SA_CH) {
        const int cw = min(SA_CH, D - c0);
        __syncthreads();
        for (int f = threadIdx.x; f < 32 * (SA_CH / 4); f += SA_THREADS) {
            const int row = f / (SA_CH / 4), c4 = f % (SA_CH / 4);
            f32x4 gv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (row < T && 4 * c4 < cw) {
                gv = *reinterpret_cast<const f32x4*>(Gb + (int64_t)row * D + c0 + 4 * c4);
                vv = *reinterpret_cast<const f32x4*>(Vb + (int64_t)row * D + c0 + 4 * c4);
            }
            *reinterpret_cast<f32x4*>(al + row * SA_LD + 4 * c4) = gv;
            *reinterpret_cast<f32x4*>(vl + row * SA_LD + 4 * c4) = vv;
        }
        __syncthreads();
        if (64 * w < cw) {
            const float* ap = al + r * SA_LD + 64 * w + 32 * h;
            const float* bp = vl + r * SA_LD + 64 * w + 32 * h;
#pragma unroll
            for (int s4 = 0; s4 < 8; ++s4) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap + 4 * s4);
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(bp + 4 * s4);
#pragma unroll
                for (int i = 0; i < 4; ++i) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i], b4[i], sacc, 0, 0, 0);
            }
        }
    }
    if (w >= 4) {
#pragma unroll
        for (int e = 0; e < 16; ++e) red[((w - 4) * 16 + e) * 64 + lane] = sacc[e];
    }
    __syncthreads();
    if (w < 4) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float t = sacc[e] + red[(w * 16 + e) * 64 + lane];
            red[(w * 16 + e) * 64 + lane] = t;
        }
    }
    __syncthreads();
    // ---- softmax backward per row i (lanes of one half = j); wave w finishes registers 2w, 2w+1
#pragma unroll
    for (int ee = 0; ee < 2; ++ee) {
        const int e = 2 * w + ee;
        const float dw = red[(0 * 16 + e) * 64 + lane] + red[(1 * 16 + e) * 64 + lane] + red[(2 * 16 + e) * 64 + lane] +
                         red[(3 * 16 + e) * 64 + lane];
        const int i = crow(e, h);
        const float wv = (i < T && r < T) ? a.w[((int64_t)b * T + i) * T + r] : 0.f;
        float dot = wv * dw;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
        wl[i][r] = wv;
        gl[i][r] = wv * (dw - dot) * a.scale;
    }
    __syncthreads();
    // ---- column-parallel passes
    if (gridDim.y > 1) {                  // columns of the clip spread over gridDim.y workgroups: one column per thread
        for (int cbase = blockIdx.y * SA_CH; cbase < D; cbase += gridDim.y * SA_CH) {
            const int c = cbase + threadIdx.x;
            float acc[32];
            __syncthreads();
            sa_stage_rows(al, Gb, T, D, cbase);           // d(out) and Q columns of this block through LDS (see the forward)
            sa_stage_rows(vl, Qb, T, D, cbase);
            __syncthreads();
            if (c < D) {
#pragma unroll
                for (int j = 0; j < 32; ++j) acc[j] = 0.f;
                for (int i = 0; i < T; ++i) {
                    const float x = al[i * SA_LD + threadIdx.x];
#pragma unroll
                    for (int j = 0; j < 32; ++j) acc[j] += wl[i][j] * x;
                }
#pragma unroll
                for (int j = 0; j < 32; ++j)
                    if (j < T) a.dV[base + (int64_t)j * D + c] = acc[j];
#pragma unroll
                for (int i = 0; i < 32; ++i) acc[i] = 0.f;
                for (int j = 0; j < T; ++j) {
                    const float x = vl[j * SA_LD + threadIdx.x];
#pragma unroll
                    for (int i = 0; i < 32; ++i) acc[i] += gl[i][j] * x;
                }
#pragma unroll
                for (int i = 0; i < 32; ++i)
                    if (i < T) a.dK[base + (int64_t)i * D + c] = acc[i];
            }
            __syncthreads();
            sa_stage_rows(al, Kb, T, D, cbase);
            __syncthreads();
            if (c < D) {
#pragma unroll
                for (int j = 0; j < 32; ++j) acc[j] = 0.f;
                for (int i = 0; i < T; ++i) {
                    const float x = al[i * SA_LD + threadIdx.x];
#pragma unroll
                    for (int j = 0; j < 32; ++j) acc[j] += gl[i][j] * x;
                }
#pragma unroll
                for (int j = 0; j < 32; ++j)
                    if (j < T) a.dQ[base + (int64_t)j * D + c] = acc[j];
            }
        }
        return;
    }
    // one thread per 4 columns
    for (int c = threadIdx.x * 4; c < D; c += SA_THREADS * 4) {
        f32x4 acc[32];
        // dV[j] = sum_i w[i][j] d(out)[i]
#pragma unroll
        for (int j = 0; j < 32; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < T; ++i) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(Gb + (int64_t)i * D + c);
#pragma unroll
            for (int j = 0; j < 32; ++j) acc[j] += wl[i][j] * x;
        }
#pragma unroll
        for (int j = 0; j < 32; ++j)
            if (j < T) *reinterpret_cast<f32x4*>(a.dV + base + (int64_t)j * D + c) = acc[j];
        // dK[i] = sum_j g[i][j] Q[j]
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < T; ++j) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(Qb + (int64_t)j * D + c);
#pragma unroll
            for (int i = 0; i < 32; ++i) acc[i] += gl[i][j] * x;
        }
#pragma unroll
        for (int i = 0; i < 32; ++i)
            if (i < T) *reinterpret_cast<f32x4*>(a.dK + base + (int64_t)i * D + c) = acc[i];
        // dQ[j] = sum_i g[i][j] K[i]
#pragma unroll
        for (int j = 0; j < 32; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < T; ++i) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(Kb + (int64_t)i * D + c);
#pragma unroll
            for (int j = 0; j < 32; ++j) acc[j] += gl[i][j] * x;
        }
#pragma unroll
        for (int j = 0; j < 32; ++j)
            if (j < T) *reinterpret_cast<f32x4*>(a.dQ + base + (int64_t)j * D + c) = acc[j];
    }
}

// ------------------------------------------------------------------------------------------------ LatentPSL backward
// One workgroup per clip, P <= 8 proposals.  d(out) (P x H) -> LayerNorm + tanh backward -> du (LDS) ->
// dadj = ov du^T (wave dot products) -> softmax backward over the frames -> dov = adj du + dlg theta (written once),
// per-clip partials of dtheta = dlg^T ov and of the LayerNorm's dgamma | dbeta.
// Replaces rowln_bwd + two batched GEMMs + softmax_bwd + two more GEMMs (six launches on 26 x 8 tiles).
constexpr int PB_MAXP = 8;
__global__ __launch_bounds__(PSL_THREADS) void latent_psl_bwd_kernel(const PslBwdPack pk) {
    const dlsg_latent_psl_bwd_args& a = pk.s[blockIdx.y];
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float adjl[PSL_MAXT][PB_MAXP + 1];
    __shared__ float dlgl[PSL_MAXT][PB_MAXP + 1];      // dadj, then dlogits
    __shared__ float red[16 * 16];
    const int b = blockIdx.x;
    const int T = a.T, P = a.P, H = a.H;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t seed = a.seed + (a.seed_ptr ? *a.seed_ptr : 0ull);
    float* ovl = smem;                       // [T][H]
    float* dul = smem + T * H;               // [PB_MAXP][H]
    {
        const float* src = a.ov + (int64_t)b * T * H;
        for (int i = threadIdx.x * 4; i < T * H; i += PSL_THREADS * 4)
            *reinterpret_cast<f32x4*>(ovl + i) = *reinterpret_cast<const f32x4*>(src + i);
        for (int i = threadIdx.x; i < T * P; i += PSL_THREADS) adjl[i / P][i % P] = a.adj[(int64_t)b * T * P + i];
    }
    // ---- LayerNorm (+ dropout, tanh) backward of the P rows, statistics of all rows reduced together
    float yv[PB_MAXP][2], gv[PB_MAXP][2], xh[PB_MAXP][2];
    float sums[2 * PB_MAXP];
#pragma unroll
    for (int i = 0; i < 2 * PB_MAXP; ++i) sums[i] = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int h = threadIdx.x + c * PSL_THREADS;
        float dgam = 0.f, dbet = 0.f;
#pragma unroll
        for (int p = 0; p < PB_MAXP; ++p) {
            yv[p][c] = 0.f; gv[p][c] = 0.f; xh[p][c] = 0.f;
            if (p < P && h < H) {
                const int64_t row = (int64_t)b * P + p;
                const float y = tanhf(a.u[row * H + h]);
                const float mean = a.stats[2 * row], rstd = a.stats[2 * row + 1];
                float g = a.dout[row * H + h];
                if (a.p > 0.f) g *= drop_scale(seed, a.site, (uint64_t)row * H + h, a.p);
                const float x = (y - mean) * rstd;
                dgam += g * x;
                dbet += g;
                const float gx = g * a.gamma[h];
                yv[p][c] = y; gv[p][c] = gx; xh[p][c] = x;
                sums[2 * p] += gx;
                sums[2 * p + 1] += gx * x;
            }
        }
        if (h < H) {
            a.part[(int64_t)b * 2 * H + h] = dgam;
            a.part[(int64_t)b * 2 * H + H + h] = dbet;
        }
    }
#pragma unroll
    for (int i = 0; i < 2 * PB_MAXP; ++i) sums[i] = wave_sum(sums[i]);
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < 2 * PB_MAXP; ++i) red[16 * i + w] = sums[i];
    __syncthreads();
#pragma unroll
    for (int p = 0; p < PB_MAXP; ++p) {
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { m1 += red[16 * (2 * p) + k]; m2 += red[16 * (2 * p + 1) + k]; }
        m1 /= H; m2 /= H;
        const float rstd = p < P ? a.stats[2 * ((int64_t)b * P + p) + 1] : 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int h = threadIdx.x + c * PSL_THREADS;
            if (h < H) dul[p * H + h] = (p < P) ? rstd * (gv[p][c] - m1 - xh[p][c] * m2) * (1.f - yv[p][c] * yv[p][c]) : 0.f;
        }
    }
    __syncthreads();
    // ---- dadj[t][p] = ov[t] . du[p]: a wave keeps du[p] in registers and walks a slice of the frames
    {
        const int nw = P <= 16 ? 16 / P : 1;
        for (int task = w; task < P * nw; task += PSL_THREADS / 64) {
            const int p = task / nw, part = task % nw;
            const int t0 = (T * part) / nw, t1 = (T * (part + 1)) / nw;
            for (int t = t0; t < t1; ++t) {
                float acc = 0.f;
                for (int j = lane * 4; j < H; j += 256) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(ovl + t * H + j);
                    const f32x4 y = *reinterpret_cast<const f32x4*>(dul + p * H + j);
                    acc += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
                }
                acc = wave_sum(acc);
                if (lane == 0) dlgl[t][p] = acc;
            }
        }
    }
    __syncthreads();
    // ---- softmax backward over the frames, per proposal
    if (threadIdx.x < P) {
        const int p = threadIdx.x;
        float dot = 0.f;
        for (int t = 0; t < T; ++t) dot += adjl[t][p] * dlgl[t][p];
        for (int t = 0; t < T; ++t) dlgl[t][p] = adjl[t][p] * (dlgl[t][p] - dot);
    }
    __syncthreads();
    // ---- dov[t][h] = sum_p adj[t][p] du[p][h] + dlg[t][p] theta[p][h];  dtheta[p][h] = sum_t dlg[t][p] ov[t][h]
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int h = threadIdx.x + c * PSL_THREADS;
        if (h >= H) continue;
        float du[PB_MAXP], th[PB_MAXP], dth[PB_MAXP];
#pragma unroll
        for (int p = 0; p < PB_MAXP; ++p) {
            du[p] = p < P ? dul[p * H + h] : 0.f;
            th[p] = p < P ? a.theta[(int64_t)p * H + h] : 0.f;
            dth[p] = 0.f;
        }
        for (int t = 0; t < T; ++t) {
            const float x = ovl[t * H + h];
            float o = 0.f;
#pragma unroll
            for (int p = 0; p < PB_MAXP; ++p) {
                const float ad = p < P ? adjl[t][p] : 0.f, dl = p < P ? dlgl[t][p] : 0.f;
                o += ad * du[p] + dl * th[p];
                dth[p] += dl * x;
            }
            a.dov[((int64_t)b * T + t) * H + h] = o;
        }
#pragma unroll
        for (int p = 0; p < PB_MAXP; ++p)
            if (p < P) a.dtheta_part[((int64_t)b * P + p) * H + h] = dth[p];
    }
}

}  // namespace

extern "C" int dlsg_latent_psl_fwd_multi(const dlsg_latent_psl_args* a, int count, void* stream) {
    if (!a || count < 1 || count > DLSG_PSL_MAXMULTI) return DLSG_EINVAL;
    PslPack pk;
    for (int i = 0; i < count; ++i) {
        const dlsg_latent_psl_args& x = a[i];
        if (x.T < 1 || x.T > PSL_MAXT || x.P < 1 || x.P > PSL_MAXP || x.H < 4 || x.H > 2 * PSL_THREADS || x.H % 4) return DLSG_EINVAL;
        if (x.B != a->B || x.P != a->P || x.H != a->H) return DLSG_EINVAL;
        if ((reinterpret_cast<uintptr_t>(x.ov) | reinterpret_cast<uintptr_t>(x.theta)) & 15) return DLSG_EINVAL;
        pk.s[i] = x;
    }
    if (a->B == 0) return DLSG_OK;
    const int lds_bytes = a->P * a->H * 4;              // theta
    if (lds_bytes > 64 * 1024) return DLSG_EINVAL;
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&latent_psl_fwd_kernel<4>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&latent_psl_fwd_kernel<8>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    });
    if (a->H <= 1024)
        hipLaunchKernelGGL(latent_psl_fwd_kernel<4>, dim3(a->B, count), dim3(PSL_THREADS), lds_bytes, reinterpret_cast<hipStream_t>(stream), pk);
    else
        hipLaunchKernelGGL(latent_psl_fwd_kernel<8>, dim3(a->B, count), dim3(PSL_THREADS), lds_bytes, reinterpret_cast<hipStream_t>(stream), pk);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_latent_psl_fwd(const dlsg_latent_psl_args* a, void* stream) { return dlsg_latent_psl_fwd_multi(a, 1, stream); }

extern "C" int dlsg_sa_core_fwd(const dlsg_sa_core_args* a, void* stream) {
    if (!a || a->T < 1 || a->T > 32 || a->D < 64 || a->D % 64) return DLSG_EINVAL;
    if ((reinterpret_cast<uintptr_t>(a->K) | reinterpret_cast<uintptr_t>(a->Q) | reinterpret_cast<uintptr_t>(a->V) |
         reinterpret_cast<uintptr_t>(a->out)) & 15)
        return DLSG_EINVAL;
    if (a->B == 0) return DLSG_OK;
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sa_core_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  SA_LDS_FLOATS * 4);
    });
    hipLaunchKernelGGL(sa_core_fwd_kernel, dim3(a->B, sa_col_split(a->B, a->D)), dim3(SA_THREADS), SA_LDS_FLOATS * 4,
                       reinterpret_cast<hipStream_t>(stream), *a);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}

extern "C" int dlsg_sa_core_bwd(const dlsg_sa_core_bwd_args* a, void* stream) {
    if (!a || a->T < 1 || a->T > 32 || a->D < 64 || a->D % 64) return DLSG_EINVAL;
    if ((reinterpret_cast<uintptr_t>(a->K) | reinterpret_cast<uintptr_t>(a->Q) | reinterpret_cast<uintptr_t>(a->V) |
         reinterpret_cast<uintptr_t>(a->dout) | reinterpret_cast<uintptr_t>(a->dK) | reinterpret_cast<uintptr_t>(a->dQ) |
         reinterpret_cast<uintptr_t>(a->dV)) & 15)
        return DLSG_EINVAL;
    if (a->B == 0) return DLSG_OK;
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sa_core_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  SA_LDS_FLOATS * 4);
    });
    hipLaunchKernelGGL(sa_core_bwd_kernel, dim3(a->B, sa_col_split(a->B, a->D)), dim3(SA_THREADS), SA_LDS_FLOATS * 4,
                       reinterpret_cast<hipStream_t>(stream), *a);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}

extern "C" int dlsg_latent_psl_bwd_multi(const dlsg_latent_psl_bwd_args* a, int count, void* stream) {
    if (!a || count < 1 || count > DLSG_PSL_MAXMULTI) return DLSG_EINVAL;
    PslBwdPack pk;
    for (int i = 0; i < count; ++i) {
        const dlsg_latent_psl_bwd_args& x = a[i];
        if (x.T < 1 || x.T > PSL_MAXT || x.P < 1 || x.P > PB_MAXP || x.H < 4 || x.H > 2 * PSL_THREADS || x.H % 4) return DLSG_EINVAL;
        if (x.B != a->B || x.T != a->T || x.H != a->H) return DLSG_EINVAL;
        if (reinterpret_cast<uintptr_t>(x.ov) & 15) return DLSG_EINVAL;
        pk.s[i] = x;
    }
    if (a->B == 0) return DLSG_OK;
    const int lds_bytes = (a->T + PB_MAXP) * a->H * 4;
    if (lds_bytes > 150 * 1024) return DLSG_EINVAL;
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&latent_psl_bwd_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    });
    hipLaunchKernelGGL(latent_psl_bwd_kernel, dim3(a->B, count), dim3(PSL_THREADS), lds_bytes, reinterpret_cast<hipStream_t>(stream), pk);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_latent_psl_bwd(const dlsg_latent_psl_bwd_args* a, void* stream) { return dlsg_latent_psl_bwd_multi(a, 1, stream); }
