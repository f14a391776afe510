// Object->frame conditional graph, backward (reference models/layer.py:184-192), second generation (gfx950).
//
// Forward: z_t = sum_n P[n,t] o_n + v_t,  P = softmax_n(S),  S[n,t] = scale o_n . v_t,  o_n = obj_norm(y_n) (LayerNorm).
// Given dz (T x H) per clip:
//     dP[n,t] = o_n . dz_t                      P[n,t] = exp(S[n,t] - M_t) / L_t        (S, M, L saved by the forward)
//     dS'     = scale P (dP - c_t),             c_t = sum_n P dP = (z_t - v_t) . dz_t
//     do_n    = sum_t P[n,t] dz_t + sum_t dS'[n,t] v_t                      (rank <= 2T in the frames)
//     dv_t    = dz_t + sum_n dS'[n,t] o_n
//     dy_n    = r_n (do_n*gamma - m1_n - xh_n m2_n) (1 - y_n^2)             (LayerNorm + tanh backward, xh = (y - mu) r)
//     m1_n = mean_h(do*gamma), m2_n = mean_h(do*gamma*xh): from per-frame dot products, without forming do
//     dgamma = sum_n do*xh,  dbeta = sum_n do.
//
// The first generation (attention.hip) staged tiles through registers, normalised them into LDS and kept dv in the column-sliced
// second pass (1024 small workgroups re-reading P / dS' sixteen times): 82 + 119 us per stream at batch 64, 1.15 TB/s of
// algorithmic bytes.  Both passes now run on the forward's machinery (o2v16.hpp): one workgroup per (clip, object chunk, stream),
// 16-object tiles double-buffered in LDS by LDS-DMA, the LayerNorm folded into the products, v_mfma_f32_16x16x4_f32.
//
//   pass 1, o2v16_bwd_scores_kernel = the forward kernel with dz in place of v:
//       dP = x . (gamma*dz)^T with the folded LayerNorm, P from the saved scores, dS', the row means m1 / m2;
//       pd (NO x 64) = [P | dS'] and m12 are written for pass 2; and -- as the forward's aggregation, the dS' registers being
//       the A operand -- dv = dz + gamma*(sum_n dS' r_n x_n - sum_n dS' r_n mu_n) + beta sum_n dS' on the raw rows.
//   pass 2, o2v16_bwd_apply_kernel: do = pd . [dz ; v] (K = 64; [dz ; v] of the wave's 128 columns lives in 128 registers as
//       the B operand, pd rows come straight from global as the A operand), LayerNorm + tanh backward IN PLACE on the LDS
//       tile, rows written out with 16-byte coalesced stores; dgamma / dbeta partials per (clip, chunk) in registers.
//
// A single pass would need gamma*dz in the dP layout (64 registers), [dz ; v] in the do layout (128) and the dv accumulators
// (64) at once: over the 256 registers of an 8-wave workgroup.  y is therefore read twice (the second time mostly from the
// Infinity Cache: both passes of a launch touch 2 x 109 MB + 109 MB per stream at batch 64).
#include <cstdlib>
#include <mutex>

#include "o2v16.hpp"

using namespace dlsg;
using namespace o16;

namespace {

struct B16Pack {
    dlsg_o2v_bwd_args s[DLSG_O2V_MAXMULTI];
};

// ================================================================================================ pass 1: scores + dv
template <int H>
__global__ __launch_bounds__(O16_THREADS) void o2v16_bwd_scores_kernel(const B16Pack pk, int tiles_per_split) {
    using G = O16Geom<H>;
    const dlsg_o2v_bwd_args& a = pk.s[blockIdx.z];
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* red = smem + 2 * G::BUF;
    float* gam_l = red + G::RED;
    float* bet_l = gam_l + H;
    __shared__ float fsc[7][32];                 // per-frame scalars q0..q4 (below), M_t, 1 / L_t: read per tile, not held

    const int b = blockIdx.x, sp = blockIdx.y;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int f = lane & 15, g = lane >> 4;
    const int T = a.T, NO = a.NO;
    const int n_begin = sp * tiles_per_split * O16_TILE;
    const int n_end = min(NO, n_begin + tiles_per_split * O16_TILE);

    auto issue_tile = [&](int n0, float* dst) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int row = 2 * w + rr;
            const int n = min(n0 + row, NO - 1);
            const char* src = reinterpret_cast<const char*>(a.y + ((int64_t)b * NO + n) * H) + lane * G::VB;
            char* d = reinterpret_cast<char*>(dst + row * G::LDO);
#pragma unroll
            for (int q = 0; q < G::PPR; ++q) glds<G::VB>(src + q * 64 * G::VB, d + q * 64 * G::VB);
        }
    };
    if (n_begin < n_end) issue_tile(n_begin, smem);

    for (int j = threadIdx.x; j < H; j += O16_THREADS) { gam_l[j] = a.g_obj[j]; bet_l[j] = a.b_obj[j]; }
    if (threadIdx.x < 32) {
        const int t = min((int)threadIdx.x, T - 1);
        fsc[5][threadIdx.x] = a.ml[2 * ((int64_t)b * T + t)];
        fsc[6][threadIdx.x] = 1.f / a.ml[2 * ((int64_t)b * T + t) + 1];
    }

    // ---- dz fragments (B operand of the dP product), gamma folded in; per-frame scalars
    //      q0 = gamma.dz_t (a_t), q1 = beta.dz_t (e_t), q2 = (z_t - v_t).dz_t (c_t), q3 = gamma.v_t (b_t), q4 = beta.v_t (f_t)
    float vreg[2][G::HS / 4];
    {
        float fq[2][5];
        const int k0 = (w % G::KW) * G::HS + 4 * g;
#pragma unroll
        for (int fb = 0; fb < 2; ++fb) {
            const int t = 16 * fb + f;
            const int64_t off = ((int64_t)b * T + min(t, T - 1)) * H + k0;
#pragma unroll
            for (int q = 0; q < 5; ++q) fq[fb][q] = 0.f;
#pragma unroll
            for (int c = 0; c < G::NCHUNK; ++c) {
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(a.dz + off + 16 * c);
                const f32x4 v4 = *reinterpret_cast<const f32x4*>(a.v + off + 16 * c);
                const f32x4 z4 = *reinterpret_cast<const f32x4*>(a.z + off + 16 * c);
                const f32x4 g4 = *reinterpret_cast<const f32x4*>(a.g_obj + k0 + 16 * c);
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.b_obj + k0 + 16 * c);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const bool on = t < T && w < G::KW;
                    const float dd = on ? d4[s] : 0.f, vv = on ? v4[s] : 0.f, zz = on ? z4[s] : 0.f;
                    vreg[fb][4 * c + s] = dd * g4[s];
                    fq[fb][0] += dd * g4[s];
                    fq[fb][1] += dd * b4[s];
                    fq[fb][2] += (zz - vv) * dd;
                    fq[fb][3] += vv * g4[s];
                    fq[fb][4] += vv * b4[s];
                }
            }
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                fq[fb][q] += __shfl_xor(fq[fb][q], 16, 64);
                fq[fb][q] += __shfl_xor(fq[fb][q], 32, 64);
                if (g == 0) red[((w * 2 + fb) * 5 + q) * 16 + f] = fq[fb][q];
            }
        }
    }
    f32x4 acc_o[2][G::CBW];
#pragma unroll
    for (int fb = 0; fb < 2; ++fb)
#pragma unroll
        for (int c = 0; c < G::CBW; ++c) acc_o[fb][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float l_run[2] = {0.f, 0.f}, c_run[2] = {0.f, 0.f};

    lds_barrier();
    if (threadIdx.x < 160) {                     // (q, frame) = 5 x 32 sums over the eight waves
        const int q = threadIdx.x >> 5, t = threadIdx.x & 31, fb = t >> 4, ff = t & 15;
        float s_ = 0.f;
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) s_ += red[((ww * 2 + fb) * 5 + q) * 16 + ff];
        fsc[q][t] = s_;
    }
    const float inv_scale = 1.f / a.scale, inv_h = 1.f / H;
    // the prologue's global loads are consumed; from here on the vector-memory queue holds LDS-DMA pieces, the small
    // statistics / score loads of a tile (consumed inside the tile) and the pd / m12 stores
    if (n_begin < n_end) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();

    int it = 0;
    for (int n0 = n_begin; n0 < n_end; n0 += O16_TILE, ++it) {
        float* cur = smem + (it & 1) * G::BUF;
        float* nxt = smem + ((it + 1) & 1) * G::BUF;
        const bool more = n0 + O16_TILE < n_end;
        if (more) issue_tile(n0 + O16_TILE, nxt);
        // ---- saved statistics and raw scores of this lane's 4 objects: in flight during the dP product
        float mu4[4], rs4[4], sval[2][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = min(n0 + 4 * g + i, NO - 1);
            const float2 st = *reinterpret_cast<const float2*>(a.ostats + 2 * ((int64_t)b * NO + n));
            mu4[i] = st.x; rs4[i] = st.y;
#pragma unroll
            for (int fb = 0; fb < 2; ++fb) sval[fb][i] = a.S[((int64_t)b * NO + n) * T + min(16 * fb + f, T - 1)];
        }
        // ---- partial dP over this wave's k slice: D[obj 4g+i][frame 16fb+f]
        f32x4 sacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        if (w < G::KW) {
            const float* ap = cur + f * G::LDO + w * G::HS + 4 * g;
#pragma unroll
            for (int c = 0; c < G::NCHUNK; ++c) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap + 16 * c);
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) {
                    sacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s_], vreg[0][4 * c + s_], sacc[0], 0, 0, 0);
                    sacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s_], vreg[1][4 * c + s_], sacc[1], 0, 0, 0);
                }
            }
            f32x4* r4 = reinterpret_cast<f32x4*>(red);
            r4[(w * 2 + 0) * 64 + lane] = sacc[0];
            r4[(w * 2 + 1) * 64 + lane] = sacc[1];
        }
        lds_barrier();
        float p[2][4];
        {
            const f32x4* r4 = reinterpret_cast<const f32x4*>(red);
            float u1[4] = {0.f, 0.f, 0.f, 0.f}, u2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int fb = 0; fb < 2; ++fb) {
                f32x4 sv = r4[fb * 64 + lane];
#pragma unroll
                for (int ww = 1; ww < G::KW; ++ww) sv += r4[(ww * 2 + fb) * 64 + lane];
                const int t = 16 * fb + f;
                const bool tval = t < T;
                const float q0 = fsc[0][t], q1 = fsc[1][t], q2 = fsc[2][t], q3 = fsc[3][t], q4 = fsc[4][t];
                const float Mt = fsc[5][t], iLt = fsc[6][t];
                float psum = 0.f, csum = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int n = n0 + 4 * g + i;
                    const bool valid = tval && (n < n_end);
                    const float dp = rs4[i] * (sv[i] - mu4[i] * q0) + q1;
                    const float P = valid ? __expf(sval[fb][i] - Mt) * iLt : 0.f;
                    const float dS = a.scale * P * (dp - q2);
                    u1[i] += P * q0 + dS * q3;
                    u2[i] += P * (dp - q1) + dS * (sval[fb][i] * inv_scale - q4);
                    // pd rows: wave w stores object 4g + (w & 3), frame block w >> 2 (every wave holds the same values)
                    if (i == (w & 3) && fb == (w >> 2) && n < n_end) {
                        float* pdp = a.pd + ((int64_t)b * NO + n) * 64 + t;
                        pdp[0] = P;
                        pdp[32] = dS;
                    }
                    p[fb][i] = dS * rs4[i];                      // the A operand of the dv aggregation carries rstd_n
                    psum += dS;
                    csum += p[fb][i] * mu4[i];
                }
                l_run[fb] += psum;
                c_run[fb] += csum;
            }
            if (w == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float s1 = row16_sum_dpp(u1[i]), s2 = row16_sum_dpp(u2[i]);
                    const int n = n0 + 4 * g + i;
                    if (f == 0 && n < n_end) {
                        a.m12[2 * ((int64_t)b * NO + n)] = s1 * inv_h;
                        a.m12[2 * ((int64_t)b * NO + n) + 1] = s2 * inv_h;
                    }
                }
            }
        }
        // ---- dv aggregation: acc_o[fb][c] (frame 16fb+4g+i, col 16cb+f) += sum_n (dS' r)[n][frame] x[n][col]
#pragma unroll
        for (int c = 0; c < G::CBW; ++c) {
            const int cb = w * G::CBW + c;
            if (cb < G::NCB) {
                const float* bp = cur + cb * 16 + f;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float bvv = bp[(4 * g + jj) * G::LDO];
                    acc_o[0][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[0][jj], bvv, acc_o[0][c], 0, 0, 0);
                    acc_o[1][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[1][jj], bvv, acc_o[1][c], 0, 0, 0);
                }
            }
        }
        if (more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // next tile landed (this wave's own rows); stores retired
        lds_barrier();
    }

#pragma unroll
    for (int fb = 0; fb < 2; ++fb) {
        l_run[fb] += __shfl_xor(l_run[fb], 16, 64);
        l_run[fb] += __shfl_xor(l_run[fb], 32, 64);
        c_run[fb] += __shfl_xor(c_run[fb], 16, 64);
        c_run[fb] += __shfl_xor(c_run[fb], 32, 64);
    }
    float gcol[G::CBW], bcol[G::CBW];
#pragma unroll
    for (int c = 0; c < G::CBW; ++c) {
        const int cb = min(w * G::CBW + c, G::NCB - 1);
        gcol[c] = gam_l[cb * 16 + f];
        bcol[c] = bet_l[cb * 16 + f];
    }
    if (a.nsplit == 1) {
        // dv = dz + gamma (acc - c) + beta l
        float res[2][G::CBW][4];
#pragma unroll
        for (int fb = 0; fb < 2; ++fb)
#pragma unroll
            for (int c = 0; c < G::CBW; ++c) {
                const int cb = min(w * G::CBW + c, G::NCB - 1);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int t = min(16 * fb + 4 * g + i, T - 1);
                    res[fb][c][i] = a.dz[((int64_t)b * T + t) * H + cb * 16 + f];
                }
            }
#pragma unroll
        for (int fb = 0; fb < 2; ++fb) {
            float lrow[4], crow[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                lrow[i] = __shfl(l_run[fb], 4 * g + i, 64);
                crow[i] = __shfl(c_run[fb], 4 * g + i, 64);
            }
#pragma unroll
            for (int c = 0; c < G::CBW; ++c) {
                const int cb = w * G::CBW + c;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int t = 16 * fb + 4 * g + i;
                    if (t < T && cb < G::NCB)
                        a.dv[((int64_t)b * T + t) * H + cb * 16 + f] =
                            gcol[c] * (acc_o[fb][c][i] - crow[i]) + bcol[c] * lrow[i] + res[fb][c][i];
                }
            }
        }
        return;
    }
    // ---- partials for o2v_combine (attention.hip): with m = 0 and l = 1 / nsplit in every chunk the combine's
    //      sum_s exp(m_s - M) agg_s / sum_s exp(m_s - M) l_s + "v" is the plain sum of the chunks + dz
    float* wsp = a.ws + ((int64_t)b * a.nsplit + sp) * ((int64_t)T * H + 64);
#pragma unroll
    for (int fb = 0; fb < 2; ++fb) {
        float lrow[4], crow[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            lrow[i] = __shfl(l_run[fb], 4 * g + i, 64);
            crow[i] = __shfl(c_run[fb], 4 * g + i, 64);
        }
#pragma unroll
        for (int c = 0; c < G::CBW; ++c) {
            const int cb = w * G::CBW + c;
            if (cb < G::NCB) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int t = 16 * fb + 4 * g + i;
                    if (t < T) wsp[(int64_t)t * H + cb * 16 + f] = gcol[c] * (acc_o[fb][c][i] - crow[i]) + bcol[c] * lrow[i];
                }
            }
        }
        const int t = 16 * fb + f;
        if (w == 0 && g == 0 && t < T) {
            wsp[(int64_t)T * H + t] = 0.f;
            wsp[(int64_t)T * H + 32 + t] = 1.f / a.nsplit;
        }
    }
}

// ================================================================================================ pass 2: do, dy, dgamma / dbeta
// Column-parallel (nothing here contracts over H): at H = 1024 a workgroup takes HALF the columns of its (clip, chunk) -- four
// waves, 16 x 512 tiles, 2 x 33 KB of LDS -- so that TWO workgroups share a CU and run out of phase: with one 8-wave workgroup
// per CU every wave reached its MFMA block, its LayerNorm epilogue and its write-out at the same time as its SIMD partner.
template <int H, int NSL>
struct ApGeom {
    static constexpr int W = H / NSL;                          // columns of a workgroup
    static constexpr int NW = 8 / NSL;                         // waves
    static constexpr int RPW = O16_TILE / NW;                  // tile rows a wave fetches / writes out
    static constexpr int LDO = W + 4;
    static constexpr int VB = (W >= 256) ? 16 : 4;
    static constexpr int PPR = W * 4 / (64 * VB);              // LDS-DMA pieces per row slice
    static constexpr int EPL = W / 64;
    static constexpr int VEC = (EPL % 4 == 0) ? 4 : 1;
    static constexpr int NCH = EPL / VEC;
    static constexpr int NCB = W / 16;
    static constexpr int CBW = (NCB + NW - 1) / NW;
    static constexpr int BUF = O16_TILE * LDO;
};

// (at least two waves per SIMD: at H = 1024 two of these workgroups share a CU out of phase -- with the column sums of dy the
//  allocation went to 260 registers and ONE workgroup per CU, 177 us instead of 117; held at 256 it spills three words)
template <int H, int NSL>
__global__ __launch_bounds__(512 / NSL) __attribute__((amdgpu_waves_per_eu(2))) void o2v16_bwd_apply_kernel(const B16Pack pk, int tiles_per_split) {
    using G = ApGeom<H, NSL>;
    const dlsg_o2v_bwd_args& a = pk.s[blockIdx.z / NSL];
    const int col0 = (blockIdx.z % NSL) * G::W;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int b = blockIdx.x, sp = blockIdx.y;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int f = lane & 15, g = lane >> 4;
    const int T = a.T, NO = a.NO;
    const int n_begin = sp * tiles_per_split * O16_TILE;
    const int n_end = min(NO, n_begin + tiles_per_split * O16_TILE);

    auto issue_tile = [&](int n0, float* dst) {
#pragma unroll
        for (int rr = 0; rr < G::RPW; ++rr) {
            const int row = G::RPW * w + rr;
            const int n = min(n0 + row, NO - 1);
            const char* src = reinterpret_cast<const char*>(a.y + ((int64_t)b * NO + n) * H + col0) + lane * G::VB;
            char* d = reinterpret_cast<char*>(dst + row * G::LDO);
#pragma unroll
            for (int q = 0; q < G::PPR; ++q) glds<G::VB>(src + q * 64 * G::VB, d + q * 64 * G::VB);
        }
    };
    if (n_begin < n_end) issue_tile(n_begin, smem);

    // ---- [dz ; v] of this wave's column blocks as the B operand: k = 16 j + 4 g + s in [0, 64): frames of dz, then of v
    float dzv[G::CBW][16];
    float gcol[G::CBW];
#pragma unroll
    for (int c = 0; c < G::CBW; ++c) {
        const int cb = min(w * G::CBW + c, G::NCB - 1);
        gcol[c] = a.g_obj[col0 + cb * 16 + f];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int k = 16 * j + 4 * g + s, t = k & 31;
                const float* src = (k < 32 ? a.dz : a.v) + ((int64_t)b * T + min(t, T - 1)) * H + col0 + cb * 16 + f;
                dzv[c][4 * j + s] = t < T ? *src : 0.f;
            }
    }
    float gsum[G::CBW], bsum[G::CBW], ysum[G::CBW];          // obj_norm dgamma, dbeta; column sums of dy (obj_embed's bias gradient)
#pragma unroll
    for (int c = 0; c < G::CBW; ++c) { gsum[c] = 0.f; bsum[c] = 0.f; ysum[c] = 0.f; }

    // pd rows (A operand: lane (f, g) holds pd[object f][16 j + 4 g .. + 3]) and the statistics of objects 4g .. 4g+3
    f32x4 pa[4];
    float mu4[4], rs4[4], m14[4], m24[4];
    auto fetch_small = [&](int n0) {
        const int na = min(n0 + f, NO - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) pa[j] = *reinterpret_cast<const f32x4*>(a.pd + ((int64_t)b * NO + na) * 64 + 16 * j + 4 * g);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = min(n0 + 4 * g + i, NO - 1);
            const float2 st = *reinterpret_cast<const float2*>(a.ostats + 2 * ((int64_t)b * NO + n));
            const float2 mm = *reinterpret_cast<const float2*>(a.m12 + 2 * ((int64_t)b * NO + n));
            mu4[i] = st.x; rs4[i] = st.y; m14[i] = mm.x; m24[i] = mm.y;
        }
    };
    if (n_begin < n_end) {
        fetch_small(n_begin);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    lds_barrier();

    int it = 0;
    for (int n0 = n_begin; n0 < n_end; n0 += O16_TILE, ++it) {
        float* cur = smem + (it & 1) * G::BUF;
        float* nxt = smem + ((it + 1) & 1) * G::BUF;
        const bool more = n0 + O16_TILE < n_end;
        if (more) issue_tile(n0 + O16_TILE, nxt);
        // ---- do[obj 4g+i][col] = sum_k pd[obj][k] [dz ; v][k][col], then the LayerNorm + tanh backward in place
#pragma unroll
        for (int c = 0; c < G::CBW; ++c) {
            const int cb = w * G::CBW + c;
            if (cb < G::NCB) {
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[j][s], dzv[c][4 * j + s], acc, 0, 0, 0);
                float* yp = cur + cb * 16 + f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = 4 * g + i;
                    const float yv = yp[row * G::LDO];
                    const float xh = (yv - mu4[i]) * rs4[i];
                    const float dd = (n0 + row < n_end) ? acc[i] : 0.f;
                    gsum[c] += dd * xh;
                    bsum[c] += dd;
                    const float dyv = rs4[i] * (dd * gcol[c] - m14[i] - xh * m24[i]) * (1.f - yv * yv);
                    yp[row * G::LDO] = dyv;
                    ysum[c] += (n0 + row < n_end) ? dyv : 0.f;
                }
            }
        }
        if (more) fetch_small(n0 + O16_TILE);         // next tile's pd rows / statistics: in flight over the write-out
        lds_barrier();                                // the whole tile now holds dy
        // ---- write out the rows this wave fetched: 16 bytes per lane, whole row slices
#pragma unroll
        for (int rr = 0; rr < G::RPW; ++rr) {
            const int row = G::RPW * w + rr;
            const int n = n0 + row;
            if (n < n_end) {
                const float* rp = cur + row * G::LDO;
                float* dp = a.dy + ((int64_t)b * NO + n) * H + col0;
#pragma unroll
                for (int c = 0; c < G::NCH; ++c) {
                    if (G::VEC == 4) {
                        *reinterpret_cast<f32x4*>(dp + c * 256 + 4 * lane) = *reinterpret_cast<const f32x4*>(rp + c * 256 + 4 * lane);
                    } else {
                        dp[c * 64 + lane] = rp[c * 64 + lane];
                    }
                }
            }
        }
        // the DMA of the next tile and its small loads were issued BEFORE these stores and the queue retires in order: once no
        // more than the stores are outstanding, both have landed (a non-final tile has all 16 rows, so the count is exact)
        if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::RPW * G::NCH) : "memory");
        lds_barrier();                                // `cur` is free for the DMA of tile i+2; tile i+1 is complete in `nxt`
    }
    // ---- obj_norm dgamma | dbeta partials of this (clip, chunk): sum the four object groups
#pragma unroll
    for (int c = 0; c < G::CBW; ++c) {
        gsum[c] += __shfl_xor(gsum[c], 16, 64); gsum[c] += __shfl_xor(gsum[c], 32, 64);
        bsum[c] += __shfl_xor(bsum[c], 16, 64); bsum[c] += __shfl_xor(bsum[c], 32, 64);
        ysum[c] += __shfl_xor(ysum[c], 16, 64); ysum[c] += __shfl_xor(ysum[c], 32, 64);
        const int cb = w * G::CBW + c;
        if (g == 0 && cb < G::NCB) {
            float* pp = a.part + ((int64_t)b * a.nsplit + sp) * 2 * H + col0 + cb * 16 + f;
            pp[0] = gsum[c];
            pp[H] = bsum[c];
            if (a.dysum) a.dysum[((int64_t)b * a.nsplit + sp) * H + col0 + cb * 16 + f] = ysum[c];
        }
    }
}

template <int H>
int launch_t(const dlsg_o2v_bwd_args* a, int count, hipStream_t st) {
    using G = O16Geom<H>;
    static std::once_flag once;
    constexpr int NSL = H >= 1024 ? 2 : 1;
    using GA = ApGeom<H, NSL>;
    constexpr int lds1 = G::LDS_FLOATS * 4, lds2 = 2 * GA::BUF * 4;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&o2v16_bwd_scores_kernel<H>), hipFuncAttributeMaxDynamicSharedMemorySize, lds1);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&o2v16_bwd_apply_kernel<H, NSL>), hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
    });
    const int tiles = (a->NO + O16_TILE - 1) / O16_TILE;
    const int tps = (tiles + a->nsplit - 1) / a->nsplit;
    B16Pack pk;
    for (int i = 0; i < count; ++i) pk.s[i] = a[i];
    const dim3 grid(a->B, a->nsplit, count);
    hipLaunchKernelGGL((o2v16_bwd_scores_kernel<H>), grid, dim3(O16_THREADS), lds1, st, pk, tps);
    hipLaunchKernelGGL((o2v16_bwd_apply_kernel<H, NSL>), dim3(a->B, a->nsplit, count * NSL), dim3(64 * GA::NW), lds2, st, pk, tps);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? DLSG_OK : DLSG_ELAUNCH;
}

}  // namespace

// called by dlsg_o2v_bwd_multi (attention.hip), which also launches the chunk combine for dv when nsplit > 1
int dlsg_o2v16_bwd(const dlsg_o2v_bwd_args* a, int count, hipStream_t st) {
    switch (a->H) {
        case 1024: return launch_t<1024>(a, count, st);
        case 512: return launch_t<512>(a, count, st);
        case 64: return launch_t<64>(a, count, st);
        default: return DLSG_EINVAL;
    }
}
