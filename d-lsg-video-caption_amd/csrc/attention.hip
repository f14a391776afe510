// Graph-attention kernels of the D-LSG hot path (gfx950):
//   * o2v_partial / o2v_combine : fused object->frame conditional graph (reference models/layer.py:184-192).
//     One pass over the projected objects: obj_norm LayerNorm in registers while staging, scores and the
//     aggregation on the f32 matrix cores, online softmax over the object axis, flash-decoding style split
//     over object chunks so that B clips fill 256 CUs.
//   * decatt fwd / bwd : the per-word attention over the cached, pre-projected proposals
//     (reference models/sublayer.py:28-43 with K/V (and the Q / output projections) hoisted out of the word loop).
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
    static constexpr int LDS_FLOATS = O2V_TILE * LDO + 4 * 16 * 64;
};

__device__ __forceinline__ int crow(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

template <int H>
__global__ __launch_bounds__(O2V_THREADS) void o2v_partial_kernel(const dlsg_o2v_args a, int tiles_per_split) {
    using G = O2VGeom<H>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* o_lds = smem;
    float* red = smem + O2V_TILE * G::LDO;            // [4][16][64]

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

    for (int n0 = n_begin; n0 < n_end; n0 += O2V_TILE) {
        // ---- stage 4 rows per wave: global -> registers -> LayerNorm -> LDS
#pragma unroll 1
        for (int rr = 0; rr < 4; ++rr) {
            const int row = 4 * w + rr;
            const int n = n0 + row;
            float x[G::EPL];
            if (n < n_end) {
                const float* yp = a.y + ((int64_t)b * NO + n) * H;
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < G::NCH; ++c) {
                    if (G::VEC == 4) {
                        const f32x4 t4 = *reinterpret_cast<const f32x4*>(yp + c * 256 + 4 * lane);
                        x[4 * c] = t4[0]; x[4 * c + 1] = t4[1]; x[4 * c + 2] = t4[2]; x[4 * c + 3] = t4[3];
                    } else {
                        x[c] = yp[c * 64 + lane];
                    }
                }
#pragma unroll
                for (int i = 0; i < G::EPL; ++i) s += x[i];
                const float mean = wave_sum(s) / H;
                float q = 0.f;
#pragma unroll
                for (int i = 0; i < G::EPL; ++i) { const float d = x[i] - mean; q += d * d; }
                const float rstd = rsqrtf(wave_sum(q) / H + a.eps);
                if (a.ostats && lane == 0) {
                    a.ostats[2 * ((int64_t)b * NO + n)] = mean;
                    a.ostats[2 * ((int64_t)b * NO + n) + 1] = rstd;
                }
#pragma unroll
                for (int c = 0; c < G::NCH; ++c) {
                    if (G::VEC == 4) {
                        const int col = c * 256 + 4 * lane;
                        const f32x4 g4 = *reinterpret_cast<const f32x4*>(a.g_obj + col);
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.b_obj + col);
                        f32x4 o4;
#pragma unroll
                        for (int i = 0; i < 4; ++i) o4[i] = (x[4 * c + i] - mean) * rstd * g4[i] + b4[i];
                        *reinterpret_cast<f32x4*>(o_lds + row * G::LDO + col) = o4;
                    } else {
                        const int col = c * 64 + lane;
                        o_lds[row * G::LDO + col] = (x[c] - mean) * rstd * a.g_obj[col] + a.b_obj[col];
                    }
                }
            } else {
#pragma unroll
                for (int c = 0; c < G::NCH; ++c) {
                    if (G::VEC == 4) {
                        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                        *reinterpret_cast<f32x4*>(o_lds + row * G::LDO + c * 256 + 4 * lane) = z4;
                    } else {
                        o_lds[row * G::LDO + c * 64 + lane] = 0.f;
                    }
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
        float arow[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) arow[e] = __shfl(alpha, crow(e, h), 64);
#pragma unroll
        for (int c = 0; c < G::CBW; ++c) {
            const int cb = w * G::CBW + c;
            if (cb < G::NCB) {
#pragma unroll
                for (int e = 0; e < 16; ++e) acc_o[c][e] *= arow[e];
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
__global__ __launch_bounds__(256) void o2v_combine_kernel(const dlsg_o2v_args a) {
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
    for (int j = threadIdx.x; j < H; j += blockDim.x) {
        float s_ = 0.f;
        for (int s = 0; s < ns; ++s) {
            const float ms = base[s * stride + (int64_t)T * H + t];
            s_ += __expf(ms - M) * base[s * stride + (int64_t)t * H + j];
        }
        zp[j] = s_ * invL + vp[j];
    }
    if (threadIdx.x == 0 && a.ml) {
        a.ml[2 * ((int64_t)b * T + t)] = M;
        a.ml[2 * ((int64_t)b * T + t) + 1] = L;
    }
}

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

// ================================================================================================ decoder attention
constexpr int DA_THREADS = 256;
constexpr int DA_MAXP = 32;

__global__ __launch_bounds__(DA_THREADS) void decatt_fwd_kernel(const dlsg_decatt_args a) {
    __shared__ float sc[DA_MAXP];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int P = a.P, Q = a.Q, H = a.H;
    const float* q = a.q + (int64_t)b * a.ldq;
    for (int s = 0; s < a.nstream; ++s) {
        const float* Kp = a.Kp[s] + (int64_t)b * P * Q;
        const float* Vp = a.Vp[s] + (int64_t)b * P * H;
        __syncthreads();
        for (int p = w; p < P; p += DA_THREADS / 64) {
            float d = 0.f;
            for (int j = lane; j < Q; j += 64) d += Kp[(int64_t)p * Q + j] * q[j];
            d = wave_sum(d);
            if (lane == 0) sc[p] = d * a.scale;
        }
        __syncthreads();
        float m = -INFINITY;
        for (int p = 0; p < P; ++p) m = fmaxf(m, sc[p]);
        float l = 0.f;
        for (int p = 0; p < P; ++p) l += __expf(sc[p] - m);
        const float inv = 1.f / l;
        if (threadIdx.x < P && a.alpha) a.alpha[(int64_t)b * a.nstream * P + s * P + threadIdx.x] = __expf(sc[threadIdx.x] - m) * inv;
        float* c = a.c[s] + (int64_t)b * a.ldc;
        for (int j = threadIdx.x; j < H; j += DA_THREADS) {
            float acc = 0.f;
            for (int p = 0; p < P; ++p) acc += __expf(sc[p] - m) * inv * Vp[(int64_t)p * H + j];
            c[j] = acc;
        }
    }
}

// alpha (saved forward weights) is read from a.f.alpha.
__global__ __launch_bounds__(DA_THREADS) void decatt_bwd_kernel(const dlsg_decatt_bwd_args a) {
    __shared__ float dw[DA_MAXP];
    __shared__ float ds[DA_MAXP];
    const dlsg_decatt_args& f = a.f;
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int P = f.P, Q = f.Q, H = f.H;
    const float* q = f.q + (int64_t)b * f.ldq;
    float* dq = a.dq + (int64_t)b * a.lddq;
    for (int s = 0; s < f.nstream; ++s) {
        const float* Kp = f.Kp[s] + (int64_t)b * P * Q;
        const float* Vp = f.Vp[s] + (int64_t)b * P * H;
        const float* wgt = f.alpha + (int64_t)b * f.nstream * P + s * P;
        const float* dc = a.dc[s] + (int64_t)b * a.lddc;
        float* dKp = a.dKp[s] + (int64_t)b * P * Q;
        float* dVp = a.dVp[s] + (int64_t)b * P * H;
        __syncthreads();
        for (int p = w; p < P; p += DA_THREADS / 64) {
            float d = 0.f;
            for (int j = lane; j < H; j += 64) d += dc[j] * Vp[(int64_t)p * H + j];
            d = wave_sum(d);
            if (lane == 0) dw[p] = d + (a.dalpha ? a.dalpha[(int64_t)b * f.nstream * P + s * P + p] : 0.f);
        }
        __syncthreads();
        if (threadIdx.x < P) {
            float dot = 0.f;
            for (int p = 0; p < P; ++p) dot += wgt[p] * dw[p];
            ds[threadIdx.x] = wgt[threadIdx.x] * (dw[threadIdx.x] - dot) * f.scale;
        }
        __syncthreads();
        for (int j = threadIdx.x; j < H; j += DA_THREADS) {
            const float g = dc[j];
            for (int p = 0; p < P; ++p) dVp[(int64_t)p * H + j] += wgt[p] * g;
        }
        for (int j = threadIdx.x; j < Q; j += DA_THREADS) {
            const float qj = q[j];
            float acc = (s == 0 && !a.accum_dq) ? 0.f : dq[j];
            for (int p = 0; p < P; ++p) {
                dKp[(int64_t)p * Q + j] += ds[p] * qj;
                acc += ds[p] * Kp[(int64_t)p * Q + j];
            }
            dq[j] = acc;
        }
    }
}

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
        default: return -1;
    }
}

extern "C" int64_t dlsg_o2v_workspace_bytes(int B, int T, int H, int nsplit) {
    return (int64_t)B * nsplit * ((int64_t)T * H + 64) * 4;
}

extern "C" int dlsg_o2v_fwd(const dlsg_o2v_args* a, void* stream) {
    if (!a || a->T < 1 || a->T > 32 || a->NO < 1 || a->nsplit < 1 || a->nsplit > 64) return DLSG_EINVAL;
    if (a->ws_bytes < dlsg_o2v_workspace_bytes(a->B, a->T, a->H, a->nsplit)) return DLSG_EINVAL;
    if (a->B == 0) return DLSG_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    switch (a->H) {
        case 1024: return o2v_launch<1024>(a, st);
        case 512: return o2v_launch<512>(a, st);
        case 64: return o2v_launch<64>(a, st);
        default: return DLSG_EINVAL;   // caller falls back to the unfused GEMM + softmax path
    }
}

extern "C" int dlsg_decatt_fwd(const dlsg_decatt_args* a, void* stream) {
    if (!a || a->P < 1 || a->P > DA_MAXP || a->nstream < 1 || a->nstream > 2) return DLSG_EINVAL;
    if (a->B == 0) return DLSG_OK;
    hipLaunchKernelGGL(decatt_fwd_kernel, dim3(a->B), dim3(DA_THREADS), 0, reinterpret_cast<hipStream_t>(stream), *a);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
extern "C" int dlsg_decatt_bwd(const dlsg_decatt_bwd_args* a, void* stream) {
    if (!a || a->f.P < 1 || a->f.P > DA_MAXP || a->f.nstream < 1 || a->f.nstream > 2 || !a->f.alpha) return DLSG_EINVAL;
    if (a->f.B == 0) return DLSG_OK;
    hipLaunchKernelGGL(decatt_bwd_kernel, dim3(a->f.B), dim3(DA_THREADS), 0, reinterpret_cast<hipStream_t>(stream), *a);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
