// Graph-attention entry points of the D-LSG hot path (gfx950):
//   * dlsg_o2v_* : the fused object->frame conditional graph (reference models/layer.py:184-192) -- the tile kernels live in
//     o2v16.hip / o2v16_bwd.hip; here: argument checks, the flash-decoding style combine over object chunks (so that B clips
//     fill 256 CUs) and the struct-size table of the ABI;
//   * decatt fwd / bwd : the per-word attention over the cached, pre-projected proposals
//     (reference models/sublayer.py:28-43 with K/V (and the Q / output projections) hoisted out of the word loop).
#include <cstdlib>
#include <mutex>

#include "common.hpp"
#include "dlsg.h"

using namespace dlsg;

namespace {

// ================================================================================================ o2v: combine of the object chunks
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
    if (a->H != 1024 && a->H != 512 && a->H != 64) return DLSG_EINVAL;   // caller falls back to the unfused GEMM + softmax path
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
