// Graph-attention kernels of the D-LSG hot path (gfx950):
//   * o2v_partial / o2v_combine : fused object->frame conditional graph (reference models/layer.py:184-192).
//     One pass over the projected objects: obj_norm LayerNorm in registers while staging, scores and the
//     aggregation on the f32 matrix cores, online softmax over the object axis, flash-decoding style split
//     over object chunks so that B clips fill 256 CUs.
//   * decatt fwd / bwd : the per-word attention over the cached, pre-projected proposals
//     (reference models/sublayer.py:28-43 with K/V (and the Q / output projections) hoisted out of the word loop).
#include <cstdlib>
#include <mutex>

#include "common.hpp"
#include "dlsg.h"

using namespace dlsg;

namespace {

// ================================================================================================ o2v forward
// Workgroup = 8 waves (512 threads), one (clip, object-chunk).  Tile = 32 objects x H in LDS.
//   S-product  : wave w contracts its H/8 slice: D[obj][frame] += O[obj][k] * V[frame][k]; V lives in registers as
//                ready-made B operands; partial S tiles are summed across the 8 waves through LDS.
//   softmax    : every wave holds the full 32x32 S tile in the MFMA C layout (lane = frame, regs = objects) and
//                updates the running max / sum redundantly (no broadcast needed).
//   agg-product: D[frame][hcol] += P[obj][frame] * O[obj][hcol]; the P registers ARE the A operand (k order =
//                C-layout row order), O comes from LDS with lanes on consecutive columns.
constexpr int O2V_THREADS = 512;
constexpr int O2V_TILE = 32;

template <int H>
struct O2VGeom {
    static constexpr int LDO = H + 4;                 // LDS row stride (floats): 16-B slots advance by 1 per row
    static constexpr int HS = H / 8;                  // k slice per wave in the S product
    static constexpr int KH = HS / 2;                 // k values per lane half
    static constexpr int EPL = H / 64;                // elements per lane when a wave holds one row
    static constexpr int VEC = (EPL % 4 == 0) ? 4 : 1;
    static constexpr int NCH = EPL / VEC;
    static constexpr int NCB = H / 32;                // 32-column blocks of the aggregation output
    static constexpr int CBW = (NCB + 7) / 8;         // column blocks per wave
    static constexpr int LDS_FLOATS = O2V_TILE * LDO + 4 * 16 * 64 + 2 * H;   // tile + reduction scratch + obj_norm gamma | beta
};

__device__ __forceinline__ int crow(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

template <int H>
__global__ __launch_bounds__(O2V_THREADS) void o2v_partial_kernel(const dlsg_o2v_args a, int tiles_per_split) {
    using G = O2VGeom<H>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* o_lds = smem;
    float* red = smem + O2V_TILE * G::LDO;            // [4][16][64]
    // obj_norm gamma / beta are read by every row of every tile: from LDS, not through 32 global loads per lane and tile
    float* gam_l = red + 4 * 16 * 64;
    float* bet_l = gam_l + H;
    for (int j = threadIdx.x; j < H; j += O2V_THREADS) { gam_l[j] = a.g_obj[j]; bet_l[j] = a.b_obj[j]; }
    __syncthreads();

    const int b = blockIdx.x, sp = blockIdx.y;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int T = a.T, NO = a.NO;
    const int n_begin = sp * tiles_per_split * O2V_TILE;
    const int n_end = min(NO, n_begin + tiles_per_split * O2V_TILE);

    // ---- V fragments (B operand of the S product): element s <-> k = w*HS + h*KH + s of frame r
    float vreg[G::KH];
    {
        const float* vp = a.v + ((int64_t)b * T + r) * H + w * G::HS + h * G::KH;
#pragma unroll
        for (int s = 0; s < G::KH; ++s) vreg[s] = (r < T) ? vp[s] : 0.f;
    }

    f32x16 acc_o[G::CBW];
#pragma unroll
    for (int c = 0; c < G::CBW; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc_o[c][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // rows of the tile being staged (4 per wave): all four rows' loads are issued before the first reduction, so a tile
    // costs one memory round trip.  (Issuing them one tile ahead, under the aggregation MFMAs, was measured: the 64
    // extra live registers spill and it is slower, 1.76 vs 1.90 TB/s.)
    float x[4][G::EPL];
    auto issue_loads = [&](int n0) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int n = n0 + 4 * w + rr;
            if (n < n_end) {
                const float* yp = a.y + ((int64_t)b * NO + n) * H;
#pragma unroll
                for (int c = 0; c < G::NCH; ++c) {
                    if (G::VEC == 4) {
                        const f32x4 t4 = *reinterpret_cast<const f32x4*>(yp + c * 256 + 4 * lane);
                        x[rr][4 * c] = t4[0]; x[rr][4 * c + 1] = t4[1]; x[rr][4 * c + 2] = t4[2]; x[rr][4 * c + 3] = t4[3];
                    } else {
                        x[rr][c] = yp[c * 64 + lane];
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < G::EPL; ++i) x[rr][i] = 0.f;
            }
        }
    };
    for (int n0 = n_begin; n0 < n_end; n0 += O2V_TILE) {
        // ---- stage: global -> registers -> LayerNorm -> LDS
        issue_loads(n0);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int row = 4 * w + rr;
            const int n = n0 + row;
            const bool valid = n < n_end;
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < G::EPL; ++i) s += x[rr][i];
            const float mean = wave_sum(s) / H;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < G::EPL; ++i) { const float d = x[rr][i] - mean; q += d * d; }
            const float rstd = rsqrtf(wave_sum(q) / H + a.eps);
            if (valid && a.ostats && lane == 0) {
                a.ostats[2 * ((int64_t)b * NO + n)] = mean;
                a.ostats[2 * ((int64_t)b * NO + n) + 1] = rstd;
            }
#pragma unroll
            for (int c = 0; c < G::NCH; ++c) {
                if (G::VEC == 4) {
                    const int col = c * 256 + 4 * lane;
                    const f32x4 g4 = *reinterpret_cast<const f32x4*>(gam_l + col);
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bet_l + col);
                    f32x4 o4;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o4[i] = valid ? (x[rr][4 * c + i] - mean) * rstd * g4[i] + b4[i] : 0.f;
                    *reinterpret_cast<f32x4*>(o_lds + row * G::LDO + col) = o4;
                } else {
                    const int col = c * 64 + lane;
                    o_lds[row * G::LDO + col] = valid ? (x[rr][c] - mean) * rstd * gam_l[col] + bet_l[col] : 0.f;
                }
            }
        }
        __syncthreads();

        // ---- partial S over this wave's k slice
        f32x16 sacc;
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
        {
            const float* ap = o_lds + r * G::LDO + w * G::HS + h * G::KH;
#pragma unroll
            for (int s4 = 0; s4 < G::KH / 4; ++s4) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap + 4 * s4);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i], vreg[4 * s4 + i], sacc, 0, 0, 0);
            }
        }
        // ---- sum the 8 partial tiles: 4..7 -> LDS, 0..3 add and republish, everyone sums 4 slots
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
        float p[16];
        float tmax = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float sv = red[(0 * 16 + e) * 64 + lane] + red[(1 * 16 + e) * 64 + lane] + red[(2 * 16 + e) * 64 + lane] +
                       red[(3 * 16 + e) * 64 + lane];
            sv *= a.scale;
            const int n = n0 + crow(e, h);
            const bool valid = n < n_end;
            if (w == 0 && valid && r < T && a.S) a.S[((int64_t)b * NO + n) * T + r] = sv;
            sv = valid ? sv : -INFINITY;
            p[e] = sv;
            tmax = fmaxf(tmax, sv);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = __expf(m_run - m_new);      // m_run = -inf on the first tile -> 0
        float psum = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            p[e] = __expf(p[e] - m_new);                // invalid rows: exp(-inf) = 0
            psum += p[e];
        }
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;

        // ---- aggregation: acc_o[frame][hcol] = alpha_frame * acc_o + sum_n P[n][frame] * O[n][hcol]
        {
            float arow[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) arow[e] = __shfl(alpha, crow(e, h), 64);
#pragma unroll
            for (int c = 0; c < G::CBW; ++c)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc_o[c][e] *= arow[e];
        }
#pragma unroll
        for (int c = 0; c < G::CBW; ++c) {
            const int cb = w * G::CBW + c;
            if (cb < G::NCB) {
                const float* bp = o_lds + cb * 32 + r;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float bv = bp[crow(e, h) * G::LDO];
                    acc_o[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(p[e], bv, acc_o[c], 0, 0, 0);
                }
            }
        }
        __syncthreads();   // tile and `red` are free for the next iteration
    }

    // ---- partial results: ws[(b*nsplit+sp)] = { agg[T][H], m[32], l[32] }
    float* wsp = a.ws + ((int64_t)b * a.nsplit + sp) * ((int64_t)T * H + 64);
#pragma unroll
    for (int c = 0; c < G::CBW; ++c) {
        const int cb = w * G::CBW + c;
        if (cb < G::NCB) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int t = crow(e, h);
                if (t < T) wsp[(int64_t)t * H + cb * 32 + r] = acc_o[c][e];
            }
        }
    }
    if (w == 0 && h == 0 && r < T) {
        wsp[(int64_t)T * H + r] = m_run;
        wsp[(int64_t)T * H + 32 + r] = l_run;
    }
}

// z[b,t,:] = sum_s exp(m_s - M) agg_s[t,:] / L + v[b,t,:]
__global__ __launch_bounds__(256) void o2v_combine_kernel(const dlsg_o2v_args a);
__device__ __forceinline__ void o2v_combine_body(const dlsg_o2v_args& a) {
    const int b = blockIdx.x, t = blockIdx.y;
    const int T = a.T, H = a.H, ns = a.nsplit;
    const int64_t stride = (int64_t)T * H + 64;
    const float* base = a.ws + (int64_t)b * ns * stride;
    float M = -INFINITY;
    for (int s = 0; s < ns; ++s) M = fmaxf(M, base[s * stride + (int64_t)T * H + t]);
    float L = 0.f;
    for (int s = 0; s < ns; ++s) {
        const float ms = base[s * stride + (int64_t)T * H + t];
        L += __expf(ms - M) * base[s * stride + (int64_t)T * H + 32 + t];
    }
    const float invL = 1.f / L;
    const float* vp = a.v + ((int64_t)b * T + t) * H;
    float* zp = a.z + ((int64_t)b * T + t) * H;
    // chunk weights once per workgroup; rows in 16-byte pieces with every chunk's load issued before the adds
    __shared__ float wsh[64];
    if (threadIdx.x < ns) wsh[threadIdx.x] = __expf(base[threadIdx.x * stride + (int64_t)T * H + t] - M) * invL;
    __syncthreads();
    if ((H & 3) == 0 && ((reinterpret_cast<uintptr_t>(vp) | reinterpret_cast<uintptr_t>(zp) | reinterpret_cast<uintptr_t>(a.ws)) & 15) == 0) {
        for (int j = threadIdx.x * 4; j < H; j += blockDim.x * 4) {
            f32x4 acc = *reinterpret_cast<const f32x4*>(vp + j);
            int s0 = 0;
            for (; s0 + 2 <= ns; s0 += 2) {
                const f32x4 p0 = *reinterpret_cast<const f32x4*>(base + s0 * stride + (int64_t)t * H + j);
                const f32x4 p1 = *reinterpret_cast<const f32x4*>(base + (s0 + 1) * stride + (int64_t)t * H + j);
                acc += wsh[s0] * p0;
                acc += wsh[s0 + 1] * p1;
            }
            if (s0 < ns) acc += wsh[s0] * *reinterpret_cast<const f32x4*>(base + s0 * stride + (int64_t)t * H + j);
            *reinterpret_cast<f32x4*>(zp + j) = acc;
        }
    } else {
        for (int j = threadIdx.x; j < H; j += blockDim.x) {
            float s_ = vp[j];
            for (int s = 0; s < ns; ++s) s_ += wsh[s] * base[s * stride + (int64_t)t * H + j];
            zp[j] = s_;
        }
    }
    if (threadIdx.x == 0 && a.ml) {
        a.ml[2 * ((int64_t)b * T + t)] = M;
        a.ml[2 * ((int64_t)b * T + t) + 1] = L;
    }
}
__global__ __launch_bounds__(256) void o2v_combine_kernel(const dlsg_o2v_args a) { o2v_combine_body(a); }

// the combine of several graphs of one shape in one launch (blockIdx.z picks the argument block)
struct O2VCombinePack {
    dlsg_o2v_args s[DLSG_O2V_MAXMULTI];
};
__device__ __forceinline__ void o2v_combine_body(const dlsg_o2v_args& a);
__global__ __launch_bounds__(256) void o2v_combine_multi_kernel(const O2VCombinePack pk) { o2v_combine_body(pk.s[blockIdx.z]); }

template <int H>
int o2v_launch(const dlsg_o2v_args* a, hipStream_t st) {
    using G = O2VGeom<H>;
    static std::once_flag once;
    constexpr int lds_bytes = G::LDS_FLOATS * 4;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&o2v_partial_kernel<H>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    });
    const int tiles = (a->NO + O2V_TILE - 1) / O2V_TILE;
    const int tps = (tiles + a->nsplit - 1) / a->nsplit;
    hipLaunchKernelGGL((o2v_partial_kernel<H>), dim3(a->B, a->nsplit), dim3(O2V_THREADS), lds_bytes, st, *a, tps);
    hipLaunchKernelGGL(o2v_combine_kernel, dim3(a->B, a->T), dim3(256), 0, st, *a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? DLSG_OK : DLSG_ELAUNCH;
}

// ================================================================================================ o2v backward
// Backward of the fused graph (z = softmax_n(scale * o v^T)^T o + v,  o = obj_norm(y)) in two passes over y instead of
// the unfused chain (recompute o, softmax, 4 batched GEMMs, softmax backward, LayerNorm backward: ~10 launches and six
// (B*NO x H) round trips per stream):
//
//   scores pass (o2v_bwd_scores_kernel, same tiling as the forward): dP = o dz^T on the matrix cores, then per object
//     row n and frame t
//        P  = exp(S - M_t) / L_t                       (S, M, L saved by the forward)
//        dS = P * (dP - c_t),  c_t = sum_n P dP = (z_t - v_t) . dz_t      (no pass over the objects needed)
//     and, because do = P dz + scale dS v is linear in dz and v, the two row means the LayerNorm backward needs follow
//     from per-frame dot products without ever forming do:
//        m1_n = mean_h(do * gamma)      = (sum_t P a_t + sum_t dS' b_t) / H,          a_t = dz_t . gamma, b_t = v_t . gamma
//        m2_n = mean_h(do * gamma * xh) = (sum_t P (dP - e_t) + sum_t dS' (S/scale - f_t)) / H,  e_t = beta . dz_t, f_t = beta . v_t
//     (dS' = scale * dS).  Output: pd (B, NO, 64) = [P (32 frame slots) | dS' (32 slots)] and m12 (B, NO, 2).
//
//   apply pass (o2v_bwd_apply_kernel, one workgroup per (clip, 64-column slice), all objects of the clip): do = pd . [dz ; v]
//     (K = 64) on the matrix cores, LayerNorm + tanh backward as the epilogue (dy written once), dv += dS'^T o and the
//     per-clip dgamma / dbeta partials accumulated in registers.  No cross-workgroup reduction anywhere.
template <int H>
__global__ __launch_bounds__(O2V_THREADS) void o2v_bwd_scores_kernel(const dlsg_o2v_bwd_args a, int tiles_per_split) {
    using G = O2VGeom<H>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float fs[5][32];
    float* o_lds = smem;
    float* red = smem + O2V_TILE * G::LDO;            // [4][16][64]
    float* gam_l = red + 4 * 16 * 64;
    float* bet_l = gam_l + H;
    for (int j = threadIdx.x; j < H; j += O2V_THREADS) { gam_l[j] = a.g_obj[j]; bet_l[j] = a.b_obj[j]; }
    __syncthreads();

    const int b = blockIdx.x, sp = blockIdx.y;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int T = a.T, NO = a.NO;
    const int n_begin = sp * tiles_per_split * O2V_TILE;
    const int n_end = min(NO, n_begin + tiles_per_split * O2V_TILE);

    // ---- per-frame scalars c, a, b, e, f (wave w: frames w, w+8, ...)
    for (int t = w; t < 32; t += 8) {
        float c5[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        if (t < T) {
            const float* dzp = a.dz + ((int64_t)b * T + t) * H;
            const float* vp = a.v + ((int64_t)b * T + t) * H;
            const float* zp = a.z + ((int64_t)b * T + t) * H;
            for (int j = lane; j < H; j += 64) {
                const float dzv = dzp[j], vv = vp[j], gg = a.g_obj[j], bb = a.b_obj[j];
                c5[0] += (zp[j] - vv) * dzv; c5[1] += dzv * gg; c5[2] += vv * gg; c5[3] += bb * dzv; c5[4] += bb * vv;
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) c5[i] = wave_sum(c5[i]);
        }
        if (lane == 0)
#pragma unroll
            for (int i = 0; i < 5; ++i) fs[i][t] = c5[i];
    }
    // ---- dz fragments (B operand of the dP product): element s <-> k = w*HS + h*KH + s of frame r
    float dzreg[G::KH];
    {
        const float* dp = a.dz + ((int64_t)b * T + r) * H + w * G::HS + h * G::KH;
#pragma unroll
        for (int s = 0; s < G::KH; ++s) dzreg[s] = (r < T) ? dp[s] : 0.f;
    }
    const float Mr = (r < T) ? a.ml[2 * ((int64_t)b * T + r)] : 0.f;
    const float Lr = (r < T) ? a.ml[2 * ((int64_t)b * T + r) + 1] : 1.f;
    const float inv_scale = 1.f / a.scale, inv_h = 1.f / H;
    __syncthreads();
    const float c_t = fs[0][r], a_t = fs[1][r], b_t = fs[2][r], e_t = fs[3][r], f_t = fs[4][r];

    float x[4][G::EPL];
    for (int n0 = n_begin; n0 < n_end; n0 += O2V_TILE) {
        // ---- stage: y -> obj_norm with the saved statistics -> LDS
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int n = n0 + 4 * w + rr;
            if (n < n_end) {
                const float* yp = a.y + ((int64_t)b * NO + n) * H;
#pragma unroll
                for (int c = 0; c < G::NCH; ++c) {
                    if (G::VEC == 4) {
                        const f32x4 t4 = *reinterpret_cast<const f32x4*>(yp + c * 256 + 4 * lane);
                        x[rr][4 * c] = t4[0]; x[rr][4 * c + 1] = t4[1]; x[rr][4 * c + 2] = t4[2]; x[rr][4 * c + 3] = t4[3];
                    } else {
                        x[rr][c] = yp[c * 64 + lane];
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < G::EPL; ++i) x[rr][i] = 0.f;
            }
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int row = 4 * w + rr;
            const int n = n0 + row;
            const bool valid = n < n_end;
            const float mean = valid ? a.ostats[2 * ((int64_t)b * NO + n)] : 0.f;
            const float rstd = valid ? a.ostats[2 * ((int64_t)b * NO + n) + 1] : 0.f;
#pragma unroll
            for (int c = 0; c < G::NCH; ++c) {
                if (G::VEC == 4) {
                    const int col = c * 256 + 4 * lane;
                    const f32x4 g4 = *reinterpret_cast<const f32x4*>(gam_l + col);
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bet_l + col);
                    f32x4 o4;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o4[i] = valid ? (x[rr][4 * c + i] - mean) * rstd * g4[i] + b4[i] : 0.f;
                    *reinterpret_cast<f32x4*>(o_lds + row * G::LDO + col) = o4;
                } else {
                    const int col = c * 64 + lane;
                    o_lds[row * G::LDO + col] = valid ? (x[rr][c] - mean) * rstd * gam_l[col] + bet_l[col] : 0.f;
                }
            }
        }
        __syncthreads();
        // ---- partial dP over this wave's k slice, summed across the 8 waves through LDS (as the forward's S product)
        f32x16 sacc;
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
        {
            const float* ap = o_lds + r * G::LDO + w * G::HS + h * G::KH;
#pragma unroll
            for (int s4 = 0; s4 < G::KH / 4; ++s4) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap + 4 * s4);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i], dzreg[4 * s4 + i], sacc, 0, 0, 0);
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
        // ---- wave w finishes C-layout registers 2w, 2w+1: 4 object rows (2 per lane half) x 32 frames
#pragma unroll
        for (int ee = 0; ee < 2; ++ee) {
            const int e = 2 * w + ee;
            const float dp = red[(0 * 16 + e) * 64 + lane] + red[(1 * 16 + e) * 64 + lane] + red[(2 * 16 + e) * 64 + lane] +
                             red[(3 * 16 + e) * 64 + lane];
            const int n = n0 + crow(e, h);
            const bool rowv = n < n_end;
            const bool valid = rowv && r < T;
            const float sval = valid ? a.S[((int64_t)b * NO + n) * T + r] : 0.f;
            const float P = valid ? __expf(sval - Mr) / Lr : 0.f;
            const float dSs = P * (dp - c_t) * a.scale;
            float u1 = P * a_t + dSs * b_t;
            float u2 = P * (dp - e_t) + dSs * (sval * inv_scale - f_t);
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) { u1 += __shfl_xor(u1, o, 64); u2 += __shfl_xor(u2, o, 64); }
            if (rowv) {
                float* pdp = a.pd + ((int64_t)b * NO + n) * 64;
                pdp[r] = P;
                pdp[32 + r] = dSs;
                if (r == 0) {
                    a.m12[2 * ((int64_t)b * NO + n)] = u1 * inv_h;
                    a.m12[2 * ((int64_t)b * NO + n) + 1] = u2 * inv_h;
                }
            }
        }
        __syncthreads();   // tile and `red` are free for the next iteration
    }
}

constexpr int AP_THREADS = 256;
constexpr int AP_LD = 68;            // LDS row stride (floats) of the 64-wide tiles
__global__ __launch_bounds__(AP_THREADS) void o2v_bwd_apply_kernel(const dlsg_o2v_bwd_args a) {
    __shared__ __attribute__((aligned(16))) float yl[64 * AP_LD];     // y tile: 64 objects x 64 columns
    __shared__ __attribute__((aligned(16))) float dzv[64 * AP_LD];    // rows 0..31: dz[t] slice, 32..63: v[t] slice
    __shared__ __attribute__((aligned(16))) float pdl[64 * AP_LD];    // pd tile: 64 objects x [P | dS']
    __shared__ float sm[64 * 4];                                      // mean, rstd, m1, m2 per object of the tile
    __shared__ float xch[2][18][64];                                  // pair exchange between waves w and w^2
    const int b = blockIdx.x, h0 = blockIdx.y * 64;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int T = a.T, NO = a.NO, H = a.H;
    for (int f = threadIdx.x; f < 64 * 16; f += AP_THREADS) {
        const int row = f >> 4, c4 = f & 15, t = row & 31;
        const float* src = (row < 32 ? a.dz : a.v) + ((int64_t)b * T + t) * H + h0 + 4 * c4;
        f32x4 val = {0.f, 0.f, 0.f, 0.f};
        if (t < T) val = *reinterpret_cast<const f32x4*>(src);
        *reinterpret_cast<f32x4*>(dzv + row * AP_LD + 4 * c4) = val;
    }
    const int mb = w >> 1, cb = w & 1;          // do product: 32-object block, 32-column block; dv product: object half = mb
    const int col = h0 + cb * 32 + r;
    const float gam = a.g_obj[col], bet = a.b_obj[col];
    float gsum = 0.f, bsum = 0.f;
    f32x16 accv;
#pragma unroll
    for (int e = 0; e < 16; ++e) accv[e] = 0.f;

    // next tile's y / pd rows travel in registers while the current tile is on the matrix cores
    f32x4 yq[4], pq[4];
    auto fetch = [&](int n0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = threadIdx.x + AP_THREADS * j;
            const int row = f >> 4, c4 = f & 15, n = n0 + row;
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            yq[j] = zero; pq[j] = zero;
            if (n < NO) {
                yq[j] = *reinterpret_cast<const f32x4*>(a.y + ((int64_t)b * NO + n) * H + h0 + 4 * c4);
                pq[j] = *reinterpret_cast<const f32x4*>(a.pd + ((int64_t)b * NO + n) * 64 + 4 * c4);
            }
        }
    };
    float st4[4] = {0.f, 0.f, 0.f, 0.f};
    auto fetch_stats = [&](int n0) {
        if (threadIdx.x < 64) {
            const int n = n0 + threadIdx.x;
            const bool v = n < NO;
            st4[0] = v ? a.ostats[2 * ((int64_t)b * NO + n)] : 0.f;
            st4[1] = v ? a.ostats[2 * ((int64_t)b * NO + n) + 1] : 0.f;
            st4[2] = v ? a.m12[2 * ((int64_t)b * NO + n)] : 0.f;
            st4[3] = v ? a.m12[2 * ((int64_t)b * NO + n) + 1] : 0.f;
        }
    };
    fetch(0);
    fetch_stats(0);
    for (int n0 = 0; n0 < NO; n0 += 64) {
        __syncthreads();                        // previous tile fully consumed (also orders the dzv fill on the first pass)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = threadIdx.x + AP_THREADS * j;
            const int row = f >> 4, c4 = f & 15;
            *reinterpret_cast<f32x4*>(yl + row * AP_LD + 4 * c4) = yq[j];
            *reinterpret_cast<f32x4*>(pdl + row * AP_LD + 4 * c4) = pq[j];
        }
        if (threadIdx.x < 64) {
#pragma unroll
            for (int i = 0; i < 4; ++i) sm[threadIdx.x * 4 + i] = st4[i];
        }
        __syncthreads();
        if (n0 + 64 < NO) { fetch(n0 + 64); fetch_stats(n0 + 64); }
        // ---- do[n][col] = sum_k pd[n][k] * dzv[k][col]   (lane half h owns k in [32h, 32h+32))
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        {
            const float* ap = pdl + (mb * 32 + r) * AP_LD + 32 * h;
            const float* bp = dzv + (32 * h) * AP_LD + cb * 32 + r;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap + 4 * q);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i], bp[(4 * q + i) * AP_LD], acc, 0, 0, 0);
            }
        }
        // ---- epilogue: obj_norm LayerNorm backward + tanh backward, dgamma / dbeta partials
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = mb * 32 + crow(e, h);
            const int n = n0 + row;
            if (n < NO) {
                const float yv = yl[row * AP_LD + cb * 32 + r];
                const float mean = sm[row * 4], rstd = sm[row * 4 + 1], m1 = sm[row * 4 + 2], m2 = sm[row * 4 + 3];
                const float xh = (yv - mean) * rstd;
                const float d = acc[e];
                gsum += d * xh;
                bsum += d;
                a.dy[((int64_t)b * NO + n) * H + col] = rstd * (d * gam - m1 - xh * m2) * (1.f - yv * yv);
            }
        }
        // ---- dv[t][col] += sum_n dS'[n][t] * o[n][col] over this wave's object half (lane half h: 16 objects)
        {
            const int nb = 32 * mb + 16 * h;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int nl = nb + s;
                const float av = pdl[nl * AP_LD + 32 + r];
                const float ov = (yl[nl * AP_LD + cb * 32 + r] - sm[nl * 4]) * sm[nl * 4 + 1] * gam + bet;
                accv = __builtin_amdgcn_mfma_f32_32x32x2f32(av, ov, accv, 0, 0, 0);
            }
        }
    }
    // ---- combine the wave pairs (w, w^2): same columns, other object block
    gsum += __shfl_xor(gsum, 32, 64);
    bsum += __shfl_xor(bsum, 32, 64);
    __syncthreads();
    if (mb == 1) {
#pragma unroll
        for (int e = 0; e < 16; ++e) xch[cb][e][lane] = accv[e];
        xch[cb][16][lane] = gsum;
        xch[cb][17][lane] = bsum;
    }
    __syncthreads();
    if (mb == 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int t = crow(e, h);
            if (t < T)
                a.dv[((int64_t)b * T + t) * H + col] = accv[e] + xch[cb][e][lane] + dzv[t * AP_LD + cb * 32 + r];
        }
        if (h == 0) {
            a.part[(int64_t)b * 2 * H + col] = gsum + xch[cb][16][lane];
            a.part[(int64_t)b * 2 * H + H + col] = bsum + xch[cb][17][lane];
        }
    }
}

template <int H>
int o2v_bwd_launch(const dlsg_o2v_bwd_args* a, hipStream_t st) {
    using G = O2VGeom<H>;
    static std::once_flag once;
    constexpr int lds_bytes = G::LDS_FLOATS * 4;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&o2v_bwd_scores_kernel<H>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    });
    const int tiles = (a->NO + O2V_TILE - 1) / O2V_TILE;
    const int tps = (tiles + a->nsplit - 1) / a->nsplit;
    hipLaunchKernelGGL((o2v_bwd_scores_kernel<H>), dim3(a->B, a->nsplit), dim3(O2V_THREADS), lds_bytes, st, *a, tps);
    hipLaunchKernelGGL(o2v_bwd_apply_kernel, dim3(a->B, a->H / 64), dim3(AP_THREADS), 0, st, *a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? DLSG_OK : DLSG_ELAUNCH;
}

// ================================================================================================ decoder attention
// One workgroup per clip: 2 x 256 threads, each half owns one attention stream (proposal set).  K' (P x Q) and
// V' (P x H) of the clip are streamed once with 16-B loads (this is the "decoder attention over cached K,V" traffic of
// 2*P*(Q+H)*4 B per clip, stream and step); dots are wave reductions, softmax over P <= 32 in LDS.
constexpr int DA_HALF = 256;
constexpr int DA_THREADS = 2 * DA_HALF;
constexpr int DA_MAXP = 72;
constexpr int DA_MAXQ = 2048;

template <bool VEC>
__device__ __forceinline__ float dot_row(const float* __restrict__ a, const float* __restrict__ b, int n, int lane) {
    float d = 0.f;
    if (VEC) {
        for (int j = 4 * lane; j < n; j += 256) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(a + j);
            const f32x4 y = *reinterpret_cast<const f32x4*>(b + j);
            d += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
        }
    } else {
        for (int j = lane; j < n; j += 64) d += a[j] * b[j];
    }
    return wave_sum(d);
}

template <bool VEC>
__global__ __launch_bounds__(DA_THREADS) void decatt_fwd_kernel(const dlsg_decatt_args a) {
    __shared__ float sc[2][DA_MAXP];
    const int b = blockIdx.x;
    const int s = threadIdx.x / DA_HALF, tid = threadIdx.x % DA_HALF;
    const int lane = tid & 63, w = tid >> 6;
    const int P = a.P, Q = a.Q, H = a.H;
    const bool active = s < a.nstream;
    const float* q = a.q + (int64_t)b * a.ldq;
    const float* Kp = active ? a.Kp[s] + (int64_t)b * P * Q : nullptr;
    const float* Vp = active ? a.Vp[s] + (int64_t)b * P * H : nullptr;
    if (active)
        for (int p = w; p < P; p += DA_HALF / 64) {
            const float d = dot_row<VEC>(Kp + (int64_t)p * Q, q, Q, lane);
            if (lane == 0) sc[s][p] = d * a.scale;
        }
    __syncthreads();
    if (!active) return;
    float m = -INFINITY;
    for (int p = 0; p < P; ++p) m = fmaxf(m, sc[s][p]);
    float l = 0.f;
    for (int p = 0; p < P; ++p) l += __expf(sc[s][p] - m);
    const float inv = 1.f / l;
    if (tid < P && a.alpha) a.alpha[(int64_t)b * a.nstream * P + s * P + tid] = __expf(sc[s][tid] - m) * inv;
    float* c = a.c[s] + (int64_t)b * a.ldc;
    if (VEC) {
        for (int j = 4 * tid; j < H; j += 4 * DA_HALF) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int p = 0; p < P; ++p) {
                const float wp = __expf(sc[s][p] - m) * inv;
                const f32x4 v = *reinterpret_cast<const f32x4*>(Vp + (int64_t)p * H + j);
                acc[0] += wp * v[0]; acc[1] += wp * v[1]; acc[2] += wp * v[2]; acc[3] += wp * v[3];
            }
            *reinterpret_cast<f32x4*>(c + j) = acc;
        }
    } else {
        for (int j = tid; j < H; j += DA_HALF) {
            float acc = 0.f;
            for (int p = 0; p < P; ++p) acc += __expf(sc[s][p] - m) * inv * Vp[(int64_t)p * H + j];
            c[j] = acc;
        }
    }
}

// alpha (saved forward weights) is read from a.f.alpha.  dK', dV' accumulate over the word loop; dq of the two streams
// is combined through LDS in a fixed order (deterministic).
template <bool VEC>
__global__ __launch_bounds__(DA_THREADS) void decatt_bwd_kernel(const dlsg_decatt_bwd_args a) {
    __shared__ float dw[2][DA_MAXP];
    __shared__ float ds[2][DA_MAXP];
    __shared__ float dqs[2][DA_MAXQ];
    const dlsg_decatt_args& f = a.f;
    const int b = blockIdx.x;
    const int s = threadIdx.x / DA_HALF, tid = threadIdx.x % DA_HALF;
    const int lane = tid & 63, w = tid >> 6;
    const int P = f.P, Q = f.Q, H = f.H;
    const bool active = s < f.nstream;
    const float* q = f.q + (int64_t)b * f.ldq;
    const float* Kp = active ? f.Kp[s] + (int64_t)b * P * Q : nullptr;
    const float* Vp = active ? f.Vp[s] + (int64_t)b * P * H : nullptr;
    const float* wgt = active ? f.alpha + (int64_t)b * f.nstream * P + s * P : nullptr;
    const float* dc = active ? a.dc[s] + (int64_t)b * a.lddc : nullptr;
    float* dKp = active ? a.dKp[s] + (int64_t)b * P * Q : nullptr;
    float* dVp = active ? a.dVp[s] + (int64_t)b * P * H : nullptr;
    if (active)
        for (int p = w; p < P; p += DA_HALF / 64) {
            const float d = dot_row<VEC>(Vp + (int64_t)p * H, dc, H, lane);
            if (lane == 0) dw[s][p] = d + (a.dalpha ? a.dalpha[(int64_t)b * f.nstream * P + s * P + p] : 0.f);
        }
    __syncthreads();
    if (active && tid < P) {
        float dot = 0.f;
        for (int p = 0; p < P; ++p) dot += wgt[p] * dw[s][p];
        ds[s][tid] = wgt[tid] * (dw[s][tid] - dot) * f.scale;
    }
    __syncthreads();
    if (active) {
        if (VEC) {
            for (int j = 4 * tid; j < H; j += 4 * DA_HALF) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(dc + j);
                for (int p = 0; p < P; ++p) {
                    f32x4* dv = reinterpret_cast<f32x4*>(dVp + (int64_t)p * H + j);
                    f32x4 t = *dv;
                    const float wp = wgt[p];
                    t[0] += wp * g[0]; t[1] += wp * g[1]; t[2] += wp * g[2]; t[3] += wp * g[3];
                    *dv = t;
                }
            }
            for (int j = 4 * tid; j < Q; j += 4 * DA_HALF) {
                const f32x4 qj = *reinterpret_cast<const f32x4*>(q + j);
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                for (int p = 0; p < P; ++p) {
                    const float dsp = ds[s][p];
                    f32x4* dk = reinterpret_cast<f32x4*>(dKp + (int64_t)p * Q + j);
                    f32x4 t = *dk;
                    t[0] += dsp * qj[0]; t[1] += dsp * qj[1]; t[2] += dsp * qj[2]; t[3] += dsp * qj[3];
                    *dk = t;
                    const f32x4 k = *reinterpret_cast<const f32x4*>(Kp + (int64_t)p * Q + j);
                    acc[0] += dsp * k[0]; acc[1] += dsp * k[1]; acc[2] += dsp * k[2]; acc[3] += dsp * k[3];
                }
                dqs[s][j] = acc[0]; dqs[s][j + 1] = acc[1]; dqs[s][j + 2] = acc[2]; dqs[s][j + 3] = acc[3];
            }
        } else {
            for (int j = tid; j < H; j += DA_HALF) {
                const float g = dc[j];
                for (int p = 0; p < P; ++p) dVp[(int64_t)p * H + j] += wgt[p] * g;
            }
            for (int j = tid; j < Q; j += DA_HALF) {
                const float qj = q[j];
                float acc = 0.f;
                for (int p = 0; p < P; ++p) {
                    dKp[(int64_t)p * Q + j] += ds[s][p] * qj;
                    acc += ds[s][p] * Kp[(int64_t)p * Q + j];
                }
                dqs[s][j] = acc;
            }
        }
    }
    __syncthreads();
    float* dq = a.dq + (int64_t)b * a.lddq;
    for (int j = threadIdx.x; j < Q; j += DA_THREADS) {
        float acc = a.accum_dq ? dq[j] : 0.f;
        acc += dqs[0][j];
        if (f.nstream > 1) acc += dqs[1][j];
        dq[j] = acc;
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int dlsg_struct_size(int which) {
    switch (which) {
        case 0: return (int)sizeof(dlsg_gemm_args);
        case 1: return (int)sizeof(dlsg_rowln_args);
        case 2: return (int)sizeof(dlsg_rowln_bwd_args);
        case 3: return (int)sizeof(dlsg_o2v_args);
        case 4: return (int)sizeof(dlsg_decatt_args);
        case 5: return (int)sizeof(dlsg_decatt_bwd_args);
        case 6: return (int)sizeof(dlsg_lstm_pw_args);
        case 7: return (int)sizeof(dlsg_lstm_pw_bwd_args);
        case 8: return (int)sizeof(dlsg_dec_mid_args);
        case 9: return (int)sizeof(dlsg_dec_tail_args);
        case 10: return (int)sizeof(dlsg_dec_mid_bwd_args);
        case 11: return (int)sizeof(dlsg_decatt_cache_grads_args);
        case 12: return (int)sizeof(dlsg_o2v_bwd_args);
        case 13: return (int)sizeof(dlsg_latent_psl_args);
        case 14: return (int)sizeof(dlsg_sa_core_args);
        case 15: return (int)sizeof(dlsg_beam_select_args);
        case 16: return (int)sizeof(dlsg_gather_multi_args);
        case 17: return (int)sizeof(dlsg_sa_core_bwd_args);
        case 18: return (int)sizeof(dlsg_latent_psl_bwd_args);
        case 19: return (int)sizeof(dlsg_bilstm_args);
        case 20: return (int)sizeof(dlsg_bilstm_bwd_args);
        case 21: return (int)sizeof(dlsg_colsum_desc);
        case 22: return (int)sizeof(dlsg_lstm_seq_args);
        case 23: return (int)sizeof(dlsg_cln_args);
        case 24: return (int)sizeof(dlsg_crit_sa_args);
        case 25: return (int)sizeof(dlsg_crit_pattn_args);
        case 26: return (int)sizeof(dlsg_crit_tsum_args);
        case 27: return (int)sizeof(dlsg_crit_score_args);
        case 28: return (int)sizeof(dlsg_crit_colsum_desc);
        case 29: return (int)sizeof(dlsg_crit_reduce_desc);
        default: return -1;
    }
}

extern "C" int64_t dlsg_o2v_workspace_bytes(int B, int T, int H, int nsplit) {
    return (int64_t)B * nsplit * ((int64_t)T * H + 64) * 4;
}

int dlsg_o2v16_partial(const dlsg_o2v_args* a, int count, hipStream_t st);   // o2v16.hip

extern "C" int dlsg_o2v_fwd_multi(const dlsg_o2v_args* a, int count, void* stream) {
    if (!a || count < 1 || count > DLSG_O2V_MAXMULTI) return DLSG_EINVAL;
    for (int i = 0; i < count; ++i) {
        if (a[i].T < 1 || a[i].T > 32 || a[i].NO < 1 || a[i].nsplit < 1 || a[i].nsplit > 64) return DLSG_EINVAL;
        if (a[i].ws_bytes < dlsg_o2v_workspace_bytes(a[i].B, a[i].T, a[i].H, a[i].nsplit)) return DLSG_EINVAL;
        if (a[i].B != a[0].B || a[i].T != a[0].T || a[i].NO != a[0].NO || a[i].H != a[0].H || a[i].nsplit != a[0].nsplit)
            return DLSG_EINVAL;          // one launch = one shape
    }
    if (a->B == 0) return DLSG_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    static const bool gen1 = getenv("DLSG_O2V_GEN1") != nullptr;    // A/B switch of tools/o2v_bench.py: the first-generation kernel
    if (!gen1 && (a->H == 1024 || a->H == 512 || a->H == 64)) {
        const int rc = dlsg_o2v16_partial(a, count, st);
        if (rc != DLSG_OK) return rc;
        if (a->nsplit > 1) {
            // (merging inside o2v16 by the chunk that arrives last -- arrival tickets, agent-scope release / acquire -- was built
            // and measured: 150 us instead of 78 us per 64-clip launch; one workgroup reading the other chunks' partials
            // in accumulator layout is far slower than this B x T-workgroup launch, and the fences are not free)
            O2VCombinePack pk;
            for (int i = 0; i < count; ++i) pk.s[i] = a[i];
            hipLaunchKernelGGL(o2v_combine_multi_kernel, dim3(a->B, a->T, count), dim3(256), 0, st, pk);
            DLSG_CHECK_LAUNCH();
        }
        return DLSG_OK;
    }
    for (int i = 0; i < count; ++i) {
        int rc;
        switch (a->H) {
            case 1024: rc = o2v_launch<1024>(a + i, st); break;
            case 512: rc = o2v_launch<512>(a + i, st); break;
            case 64: rc = o2v_launch<64>(a + i, st); break;
            default: return DLSG_EINVAL;   // caller falls back to the unfused GEMM + softmax path
        }
        if (rc != DLSG_OK) return rc;
    }
    return DLSG_OK;
}
extern "C" int dlsg_o2v_fwd(const dlsg_o2v_args* a, void* stream) { return dlsg_o2v_fwd_multi(a, 1, stream); }
int dlsg_o2v16_bwd(const dlsg_o2v_bwd_args* a, int count, hipStream_t st);   // o2v16_bwd.hip

extern "C" int dlsg_o2v_bwd_multi(const dlsg_o2v_bwd_args* a, int count, void* stream) {
    if (!a || count < 1 || count > DLSG_O2V_MAXMULTI) return DLSG_EINVAL;
    for (int i = 0; i < count; ++i) {
        if (a[i].T < 1 || a[i].T > 32 || a[i].NO < 1 || a[i].nsplit < 1 || a[i].nsplit > 64) return DLSG_EINVAL;
        if (a[i].B != a[0].B || a[i].T != a[0].T || a[i].NO != a[0].NO || a[i].H != a[0].H || a[i].nsplit != a[0].nsplit)
            return DLSG_EINVAL;          // one launch = one shape
        if (a[i].nsplit > 1 && (!a[i].ws || a[i].ws_bytes < dlsg_o2v_workspace_bytes(a[i].B, a[i].T, a[i].H, a[i].nsplit)))
            return DLSG_EINVAL;
    }
    if (a->B == 0) return DLSG_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    static const bool gen1 = getenv("DLSG_O2V_BWD_GEN1") != nullptr;    // A/B switch: the first-generation two-pass kernels
    if (gen1) {
        // the first generation writes one dgamma | dbeta partial per clip: rows [0, B) of `part`; the caller zero-fills the rest
        for (int i = 0; i < count; ++i) {
            int rc;
            switch (a->H) {
                case 1024: rc = o2v_bwd_launch<1024>(a + i, st); break;
                case 512: rc = o2v_bwd_launch<512>(a + i, st); break;
                case 64: rc = o2v_bwd_launch<64>(a + i, st); break;
                default: return DLSG_EINVAL;
            }
            if (rc != DLSG_OK) return rc;
        }
        return DLSG_OK;
    }
    if (a->H != 1024 && a->H != 512 && a->H != 64) return DLSG_EINVAL;
    const int rc = dlsg_o2v16_bwd(a, count, st);
    if (rc != DLSG_OK) return rc;
    if (a->nsplit > 1) {
        // dv = dz + sum over the object chunks of the partial aggregations: the forward's combine launch with m = 0, l = 1/nsplit
        O2VCombinePack pk;
        for (int i = 0; i < count; ++i) {
            dlsg_o2v_args c = {};
            c.v = a[i].dz; c.z = a[i].dv; c.ml = nullptr; c.ws = a[i].ws; c.ws_bytes = a[i].ws_bytes;
            c.B = a[i].B; c.T = a[i].T; c.NO = a[i].NO; c.H = a[i].H; c.nsplit = a[i].nsplit;
            pk.s[i] = c;
        }
        hipLaunchKernelGGL(o2v_combine_multi_kernel, dim3(a->B, a->T, count), dim3(256), 0, st, pk);
        DLSG_CHECK_LAUNCH();
    }
    return DLSG_OK;
}
extern "C" int dlsg_o2v_bwd(const dlsg_o2v_bwd_args* a, void* stream) { return dlsg_o2v_bwd_multi(a, 1, stream); }
extern "C" int dlsg_o2v_bwd_gen1(void) { return getenv("DLSG_O2V_BWD_GEN1") != nullptr; }

extern "C" int dlsg_decatt_fwd(const dlsg_decatt_args* a, void* stream) {
    if (!a || a->P < 1 || a->P > DA_MAXP || a->nstream < 1 || a->nstream > 2) return DLSG_EINVAL;
    if (a->B == 0) return DLSG_OK;
    bool vec = (a->Q % 4 == 0) && (a->H % 4 == 0) && (a->ldq % 4 == 0) && (a->ldc % 4 == 0) && aligned16(a->q);
    for (int s = 0; s < a->nstream; ++s) vec = vec && aligned16(a->Kp[s]) && aligned16(a->Vp[s]) && aligned16(a->c[s]);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (vec) hipLaunchKernelGGL(decatt_fwd_kernel<true>, dim3(a->B), dim3(DA_THREADS), 0, st, *a);
    else hipLaunchKernelGGL(decatt_fwd_kernel<false>, dim3(a->B), dim3(DA_THREADS), 0, st, *a);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_decatt_bwd(const dlsg_decatt_bwd_args* a, void* stream) {
    if (!a || a->f.P < 1 || a->f.P > DA_MAXP || a->f.nstream < 1 || a->f.nstream > 2 || !a->f.alpha) return DLSG_EINVAL;
    if (a->f.Q > DA_MAXQ) return DLSG_EINVAL;
    if (a->f.B == 0) return DLSG_OK;
    const dlsg_decatt_args& f = a->f;
    bool vec = (f.Q % 4 == 0) && (f.H % 4 == 0) && (f.ldq % 4 == 0) && (a->lddc % 4 == 0) && aligned16(f.q);
    for (int s = 0; s < f.nstream; ++s)
        vec = vec && aligned16(f.Kp[s]) && aligned16(f.Vp[s]) && aligned16(a->dc[s]) && aligned16(a->dKp[s]) && aligned16(a->dVp[s]);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (vec) hipLaunchKernelGGL(decatt_bwd_kernel<true>, dim3(f.B), dim3(DA_THREADS), 0, st, *a);
    else hipLaunchKernelGGL(decatt_bwd_kernel<false>, dim3(f.B), dim3(DA_THREADS), 0, st, *a);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
