// DiscV2's LSTM (nn.LSTM(512, 512), models/model.py:122) as persistent launches at the three differentiation levels of a WGAN-GP
// critic update (run_gun.py:362-371: the gradient penalty differentiates the critic twice).
//
// Step by step every level is one 192 x 2048 x 512 product (10-15 us) plus one cell launch per word = 78 products and 104 cell
// launches per critic update.  Here every level is ONE launch for all 26 steps, on the scheme of bilstm.hip (persist.hpp): a
// workgroup per (64-row group, 8 hidden units), its 32 x H slice of W_hh (or W_hh^T) resident in LDS, the recurrent vector exchanged
// between the workgroups through L2 with write-through stores and per-step flags.  Tensors are batch-major, as critic.py holds
// them ((n, L, 4H) pre-activations / their gradients, (n, L, H) states: a caption's words are consecutive rows), or time-major
// ((L, n, .), batch_major = 0).  n <= 256 rows = up to four row groups; the row groups are independent recurrences sharing the
// weights (the real, fake and mixed captions of one critic pass).
//
//   level 0  dlsg_lstm_seq_fwd  :  a_t = xin_t (+ b_ih + b_hh) + h_{t-1} W^T,  (h_t, c_t) = cell(a_t, c_{t-1});  Hprev_t = h_{t-1}
//   level 1  dlsg_lstm_seq_bwd  :  dh_t = dHs_t + DA_{t+1} W,  dc_t = s_t + dCs_t,  (da_t, s_{t-1}) = cell'(a_t, c_{t-1}; dh_t, dc_t),
//                                  DA_t = da_t + dAs_t           (backwards in time; DH_t = dh_t and DC_t = dc_t are kept)
//   level 2  dlsg_lstm_seq_bwd2 :  ubar_t = Ubar_t + gdh_{t-1} W^T,
//                                  (ga_t, gc_{t-1}, gdh_t, gdc_t) = cell''(a_t, c_{t-1}, DH_t, DC_t; ubar_t, gdc_{t-1})    (forwards)
//                                  gDHprev_t = gdh_{t-1}
// Cell formulas: below, with i = s(a_i), f = s(a_f), g = tanh(a_g), o = s(a_o), c = f c_prev + i g, tc = tanh(c), q = 1 - tc^2,
// s_i = i(1-i), s_f = f(1-f), s_o = o(1-o), s_g = 1 - g^2:
//   level 1:  dct = dc + dh o q;  da_i = dct g s_i;  da_f = dct c_prev s_f;  da_g = dct i s_g;  da_o = dh tc s_o;  dc_prev = dct f
//   level 2:  A  = u_i g s_i + u_f c_prev s_f + u_g i s_g + uc f            (= dL/d dct)
//             Gc = q (u_o dh s_o - 2 A dh o tc)                              (= dL/dc through tc)
//             ga_i = dct s_i (u_i g (1-2i) + u_g s_g) + Gc g s_i;   ga_f = dct s_f (u_f c_prev (1-2f) + uc) + Gc c_prev s_f
//             ga_g = dct s_g (u_i s_i - 2 u_g i g) + Gc i s_g;      ga_o = s_o dh (A q + u_o tc (1-2o))
//             gc_prev = dct u_f s_f + Gc f;   gdh = A o q + u_o tc s_o;   gdc = A
// (checked against autograd's own double backward of the unrolled cell, tests/test_gpu_ops.py).  The weight-gradient products over
// all steps are single GEMMs of the caller (critic.py): Hprev / gDHprev are their operands.
#include <mutex>

#include "dlsg.h"
#include "persist.hpp"

namespace {

using namespace persist;
using dlsg::dpp_f32;
using dlsg::sigmoidf_;

struct Cell {
    float i, f, g, o, c, tc, q;
};
__device__ __forceinline__ Cell cell_of(float ai, float af, float ag, float ao, float cp) {
    Cell x;
    x.i = sigmoidf_(ai); x.f = sigmoidf_(af); x.g = tanhf(ag); x.o = sigmoidf_(ao);
    x.c = x.f * cp + x.i * x.g;
    x.tc = tanhf(x.c);
    x.q = 1.f - x.tc * x.tc;
    return x;
}

// ================================================================================================ level 0 and level 2 (forwards in time)
// MODE 0: the forward recurrence.  MODE 2: the backward of the backward (same product h W^T, another cell).
template <int J, int MODE>
__global__ __launch_bounds__(THREADS) void lstm_seq_fwdlike_kernel(const dlsg_lstm_seq_args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int H = 64 * J, WGS = H / 8;
    f32x4* wimg = reinterpret_cast<f32x4*>(smem);
    float* red = smem + 32 * H;
    const int rg = blockIdx.x / WGS, wg = blockIdx.x % WGS, u0 = wg * 8;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int L = a.L, n = a.n;
    const bool bm = a.batch_major != 0;
    const int pstep = bm ? 1 : n;                           // rows between step t and step t - 1 of one sequence
    const bool cell_lane = c < 8;
    const int unit = u0 + (c & 7);
    const int row0 = rg * ROWS + 16 * q + 4 * w;            // this lane's 4 consecutive rows (of the whole batch)
    const int pos0 = 16 * q + 4 * w;                        // ... and their position inside the exchange slot

    fill_wimage_rows<J>(wimg, a.W, H, [&](int cb, int cc) { return (cb * 2 + (cc >> 3)) * H + u0 + (cc & 7); });
    const int RG = (n + ROWS - 1) / ROWS;
    const __amdgpu_buffer_rsrc_t xb = __builtin_amdgcn_make_buffer_rsrc(a.xbuf, 0, RG * L * H * ROWS * 4, 0x00020000);
    float st[4] = {0.f, 0.f, 0.f, 0.f};                     // MODE 0: cell state c;  MODE 2: gdc of the previous step
    float xprev[4] = {0.f, 0.f, 0.f, 0.f};                  // h_{t-1} / gdh_{t-1} of this lane's cells
    float bias[4] = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 0 && a.b_ih && cell_lane) {
#pragma unroll
        for (int g = 0; g < 4; ++g) bias[g] = a.b_ih[g * H + unit] + a.b_hh[g * H + unit];
    }
    __syncthreads();

    for (int t = 0; t < L; ++t) {
        // ---- this step's addend (xin_t / Ubar_t) and, for MODE 2, the saved tensors of the lane's cells: in flight during the wait
        float pre[4][4];
        float sa[4][4], cp[4], dh[4], dc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = row0 + i;
            const bool on = cell_lane && row < n;
            const int64_t rt = bm ? (int64_t)row * L + t : (int64_t)t * n + row;
#pragma unroll
            for (int g = 0; g < 4; ++g) pre[g][i] = on ? a.addend[rt * (4 * H) + g * H + unit] + bias[g] : 0.f;
            if (MODE == 2) {
#pragma unroll
                for (int g = 0; g < 4; ++g) sa[g][i] = on ? a.As[rt * (4 * H) + g * H + unit] : 0.f;
                cp[i] = (on && t > 0) ? a.Cs[(rt - pstep) * H + unit] : 0.f;
                dh[i] = on ? a.DH[rt * H + unit] : 0.f;
                dc[i] = on ? a.DC[rt * H + unit] : 0.f;
            }
        }
        if (t > 0) {
            const uint32_t* fl = a.flags + (int64_t)(rg * L + (t - 1)) * WGS + w * (2 * J);
            if (!wait_flags(fl, 2 * J, 1u) && lane == 0 && a.err) atomicExch(a.err, 4 + MODE);
            float own[4][2];
            product_64x32<J>(xb, ((rg * L + (t - 1)) * H) * ROWS * 4, wimg, red, own);
            // lane c < 8 holds gates i (cb 0) and g (cb 1) of its unit; f and o sit 8 lanes up in the same row of 16
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float f_ = dpp_f32<0x128>(own[i][0]);              // row_ror:8
                const float o_ = dpp_f32<0x128>(own[i][1]);
                pre[0][i] += own[i][0]; pre[1][i] += f_; pre[2][i] += own[i][1]; pre[3][i] += o_;
            }
        }
        f32x4 xv;                                           // what the next step's product contracts: h_t / gdh_t
        float o0[4][4], o1[4], o2[4];                       // per-mode outputs, stored after the publish
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (MODE == 0) {
                const Cell x = cell_of(pre[0][i], pre[1][i], pre[2][i], pre[3][i], st[i]);
                st[i] = x.c;
                xv[i] = x.o * x.tc;
            } else {
                const Cell x = cell_of(sa[0][i], sa[1][i], sa[2][i], sa[3][i], cp[i]);
                const float ui = pre[0][i], uf = pre[1][i], ug = pre[2][i], uo = pre[3][i], ucv = st[i], dhv = dh[i];
                const float si = x.i * (1.f - x.i), sf = x.f * (1.f - x.f), so = x.o * (1.f - x.o), sg = 1.f - x.g * x.g;
                const float dct = dc[i] + dhv * x.o * x.q;
                const float A = ui * x.g * si + uf * cp[i] * sf + ug * x.i * sg + ucv * x.f;
                const float Gc = x.q * (uo * dhv * so - 2.f * A * dhv * x.o * x.tc);
                o0[0][i] = dct * si * (ui * x.g * (1.f - 2.f * x.i) + ug * sg) + Gc * x.g * si;
                o0[1][i] = dct * sf * (uf * cp[i] * (1.f - 2.f * x.f) + ucv) + Gc * cp[i] * sf;
                o0[2][i] = dct * sg * (ui * si - 2.f * ug * x.i * x.g) + Gc * x.i * sg;
                o0[3][i] = so * dhv * (A * x.q + uo * x.tc * (1.f - 2.f * x.o));
                o1[i] = dct * uf * sf + Gc * x.f;           // gc_{t-1}
                xv[i] = A * x.o * x.q + uo * x.tc * so;     // gdh_t
                st[i] = A;                                  // gdc_t
                o2[i] = A;
            }
        }
        // ---- publish: write-through stores, drained by every wave, then one flag
        if (cell_lane) st_sc1(xb, (((rg * L + t) * H + unit) * ROWS + pos0) * 4, xv);
        publish(a.flags + (int64_t)(rg * L + t) * WGS + wg);
        // ---- the rest goes out with plain stores while the next step is already waiting
        if (cell_lane) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + i;
                if (row >= n) continue;
                const int64_t rt = bm ? (int64_t)row * L + t : (int64_t)t * n + row;
                if (MODE == 0) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) a.As[rt * (4 * H) + g * H + unit] = pre[g][i];
                    a.Hs[rt * H + unit] = xv[i];
                    a.Cs[rt * H + unit] = st[i];
                    if (a.Hprev) a.Hprev[rt * H + unit] = xprev[i];
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        a.gA[rt * (4 * H) + g * H + unit] = o0[g][i];
                        if (a.addend_out) a.addend_out[rt * (4 * H) + g * H + unit] = pre[g][i];   // ubar_t (the gradient w.r.t. dAs)
                    }
                    if (t > 0) a.gC[(rt - pstep) * H + unit] = o1[i];
                    if (t == L - 1) a.gC[rt * H + unit] = 0.f;                          // nothing follows the last step
                    a.gDH[rt * H + unit] = xv[i];
                    a.gDC[rt * H + unit] = o2[i];
                    if (a.gDHprev) a.gDHprev[rt * H + unit] = xprev[i];
                }
                xprev[i] = xv[i];
            }
        }
    }
}

// ================================================================================================ level 1 (backwards in time)
template <int J>
__global__ __launch_bounds__(THREADS) void lstm_seq_bwd_kernel(const dlsg_lstm_seq_args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int H = 64 * J, WGS = H / 8;
    f32x4* wimg = reinterpret_cast<f32x4*>(smem);
    float* red = smem + 32 * H;
    const int rg = blockIdx.x / WGS, wg = blockIdx.x % WGS, nn = wg >> 2, kq = wg & 3;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int L = a.L, n = a.n;
    const bool bm = a.batch_major != 0;
    const int pstep = bm ? 1 : n;
    const int pj = threadIdx.x >> 5, pp = threadIdx.x & 31, punit = 8 * wg + pj;       // pointwise cells: rows 2 pp, 2 pp + 1

    fill_wimage_cols<J>(wimg, a.W, H, kq * H, 32 * nn);
    const int RG = (n + ROWS - 1) / ROWS;
    const int xbytes = RG * L * 4 * H * ROWS * 4;
    const __amdgpu_buffer_rsrc_t gx = __builtin_amdgcn_make_buffer_rsrc(a.xbuf, 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t px = __builtin_amdgcn_make_buffer_rsrc(a.xbuf2, 0, xbytes, 0x00020000);
    uint32_t* fa = a.flags;                                 // [rg][s][wg]: DA of step s published
    uint32_t* fb = a.flags + RG * L * WGS;                  // [rg][s][wg]: partial dh of step s published
    float sdc[2] = {0.f, 0.f};
    __syncthreads();

    for (int s = 0; s < L; ++s) {
        const int t = L - 1 - s;
        float sa[4][2], cp[2], dhv[2], dcv[2], inj[4][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = rg * ROWS + 2 * pp + i;
            const bool on = row < n;
            const int64_t rt = bm ? (int64_t)row * L + t : (int64_t)t * n + row;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                sa[g][i] = on ? a.As[rt * (4 * H) + g * H + punit] : 0.f;
                inj[g][i] = (on && a.dAs) ? a.dAs[rt * (4 * H) + g * H + punit] : 0.f;
            }
            cp[i] = (on && t > 0) ? a.Cs[(rt - pstep) * H + punit] : 0.f;
            dhv[i] = on ? a.dHs[rt * H + punit] : 0.f;
            dcv[i] = sdc[i] + ((on && a.dCs) ? a.dCs[rt * H + punit] : 0.f);
        }
        if (s > 0) {
            const uint32_t* fl = fb + (int64_t)(rg * L + (s - 1)) * WGS + 4 * nn;
            if (!wait_flags(fl, 4, 1u) && lane == 0 && a.err) atomicExch(a.err, 7);
            const int pbase = ((((rg * L + (s - 1)) * 4) * H + punit) * ROWS + 2 * pp) * 4;
            f32x2 part[4];
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) part[k4] = ld2_sc1(px, pbase + k4 * H * ROWS * 4);
            dhv[0] += (part[0].x + part[1].x) + (part[2].x + part[3].x);
            dhv[1] += (part[0].y + part[1].y) + (part[2].y + part[3].y);
        }
        float da[4][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const Cell x = cell_of(sa[0][i], sa[1][i], sa[2][i], sa[3][i], cp[i]);
            const float dct = dcv[i] + dhv[i] * x.o * x.q;
            da[0][i] = dct * x.g * x.i * (1.f - x.i) + inj[0][i];
            da[1][i] = dct * cp[i] * x.f * (1.f - x.f) + inj[1][i];
            da[2][i] = dct * x.i * (1.f - x.g * x.g) + inj[2][i];
            da[3][i] = dhv[i] * x.tc * x.o * (1.f - x.o) + inj[3][i];
            sdc[i] = dct * x.f;
        }
        if (t > 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                st2_sc1(gx, ((((rg * L + s) * 4 + g) * H + punit) * ROWS + 2 * pp) * 4, f32x2{da[g][0], da[g][1]});
            publish(fa + (int64_t)(rg * L + s) * WGS + wg);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = rg * ROWS + 2 * pp + i;
            if (row >= n) continue;
            const int64_t rt = bm ? (int64_t)row * L + t : (int64_t)t * n + row;
#pragma unroll
            for (int g = 0; g < 4; ++g) a.DA[rt * (4 * H) + g * H + punit] = da[g][i];
            a.DH[rt * H + punit] = dhv[i];
            a.DC[rt * H + punit] = dcv[i];
        }
        if (t == 0) break;
        // ---- partial DA_t W over gate kq for output units 32 nn .. 32 nn + 31
        {
            const uint32_t* fl = fa + (int64_t)(rg * L + s) * WGS + w * (2 * J);
            if (!wait_flags(fl, 2 * J, 1u) && lane == 0 && a.err) atomicExch(a.err, 8);
            float own[4][2];
            product_64x32<J>(gx, (((rg * L + s) * 4 + kq) * H) * ROWS * 4, wimg, red, own);
            const int obase = ((((rg * L + s) * 4 + kq) * H + 32 * nn + c) * ROWS + 16 * q + 4 * w) * 4;
            st_sc1(px, obase, f32x4{own[0][0], own[1][0], own[2][0], own[3][0]});
            st_sc1(px, obase + 16 * ROWS * 4, f32x4{own[0][1], own[1][1], own[2][1], own[3][1]});
            publish(fb + (int64_t)(rg * L + s) * WGS + wg);
        }
    }
}

template <class K>
int launch(K kernel, int H, const dlsg_lstm_seq_args* a, hipStream_t st, std::once_flag& once) {
    const int lds_bytes = (32 * H + 4 * 3 * 8 * 64) * 4;
    std::call_once(once, [&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); });
    const int RG = (a->n + ROWS - 1) / ROWS;
    hipLaunchKernelGGL(kernel, dim3(RG * (H / 8)), dim3(THREADS), lds_bytes, st, *a);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}

}  // namespace

extern "C" int dlsg_lstm_seq_supported(int L, int n, int H) {
    if (!(L >= 1 && n >= 1 && n <= 4 * ROWS && (H == 64 || H == 512))) return 0;
    if ((int64_t)((n + ROWS - 1) / ROWS) * L * 4 * H * ROWS * 4 >= (int64_t)1 << 31) return 0;      // 32-bit buffer offsets
    return device_cus() >= ((n + ROWS - 1) / ROWS) * (H / 8) ? 1 : 0;
}
extern "C" int64_t dlsg_lstm_seq_x_floats(int L, int n, int H) { return (int64_t)((n + ROWS - 1) / ROWS) * L * 4 * H * ROWS; }
extern "C" int64_t dlsg_lstm_seq_flag_words(int L, int n, int H) { return ((int64_t)2 * ((n + ROWS - 1) / ROWS) * L * (H / 8) + 3) / 4 * 4; }

extern "C" int dlsg_lstm_seq(const dlsg_lstm_seq_args* a, int level, void* stream) {
    if (!a || level < 0 || level > 2 || !dlsg_lstm_seq_supported(a->L, a->n, a->H) || !a->W || !a->xbuf || !a->flags) return DLSG_EINVAL;
    if (level == 0 && (!a->addend || !a->As || !a->Hs || !a->Cs)) return DLSG_EINVAL;
    if (level == 1 && (!a->As || !a->Cs || !a->dHs || !a->DA || !a->DH || !a->DC || !a->xbuf2)) return DLSG_EINVAL;
    if (level == 2 && (!a->As || !a->Cs || !a->DH || !a->DC || !a->addend || !a->gA || !a->gC || !a->gDH || !a->gDC)) return DLSG_EINVAL;
    if ((a->b_ih == nullptr) != (a->b_hh == nullptr)) return DLSG_EINVAL;
    if ((reinterpret_cast<uintptr_t>(a->xbuf) & 15) || (reinterpret_cast<uintptr_t>(a->xbuf2) & 15) || (reinterpret_cast<uintptr_t>(a->W) & 15))
        return DLSG_EALIGN;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (hipMemsetAsync(a->flags, 0, dlsg_lstm_seq_flag_words(a->L, a->n, a->H) * 4, st) != hipSuccess) return DLSG_ELAUNCH;
    static std::once_flag o[6];
    if (a->H == 512) {
        if (level == 0) return launch(&lstm_seq_fwdlike_kernel<8, 0>, 512, a, st, o[0]);
        if (level == 1) return launch(&lstm_seq_bwd_kernel<8>, 512, a, st, o[1]);
        return launch(&lstm_seq_fwdlike_kernel<8, 2>, 512, a, st, o[2]);
    }
    if (level == 0) return launch(&lstm_seq_fwdlike_kernel<1, 0>, 64, a, st, o[3]);
    if (level == 1) return launch(&lstm_seq_bwd_kernel<1>, 64, a, st, o[4]);
    return launch(&lstm_seq_fwdlike_kernel<1, 2>, 64, a, st, o[5]);
}
