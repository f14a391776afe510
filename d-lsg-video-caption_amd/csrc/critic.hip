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
    const Gates x = gates_of(a, lda, r, j, H, c_prev[idx]);
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

__global__ __launch_bounds__(256) void cell_bwd2_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ c_prev,
                                                        const float* __restrict__ dh, const float* __restrict__ dc,
                                                        const float* __restrict__ u, const float* __restrict__ uc,
                                                        float* __restrict__ ga, float* __restrict__ gc_prev,
                                                        float* __restrict__ gdh, float* __restrict__ gdc, int rows, int H) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * H) return;
    const int r = idx / H, j = idx - r * H;
    const float cp = c_prev[idx];
    const Gates x = gates_of(a, lda, r, j, H, cp);
    const float dhv = dh[idx];
    const float* ur = u + (int64_t)r * 4 * H + j;
    const float ui = ur[0], uf = ur[H], ug = ur[2 * H], uo = ur[3 * H], ucv = uc[idx];
    const float si = x.i * (1.f - x.i), sf = x.f * (1.f - x.f), so = x.o * (1.f - x.o), sg = 1.f - x.g * x.g;
    const float dct = dc[idx] + dhv * x.o * x.q;
    const float A = ui * x.g * si + uf * cp * sf + ug * x.i * sg + ucv * x.f;
    const float Gc = x.q * (uo * dhv * so - 2.f * A * dhv * x.o * x.tc);
    float* g = ga + (int64_t)r * 4 * H + j;
    g[0] = dct * si * (ui * x.g * (1.f - 2.f * x.i) + ug * sg) + Gc * x.g * si;
    g[H] = dct * sf * (uf * cp * (1.f - 2.f * x.f) + ucv) + Gc * cp * sf;
    g[2 * H] = dct * sg * (ui * si - 2.f * ug * x.i * x.g) + Gc * x.i * sg;
    g[3 * H] = so * dhv * (A * x.q + uo * x.tc * (1.f - 2.f * x.o));
    gc_prev[idx] = dct * uf * sf + Gc * x.f;
    gdh[idx] = A * x.o * x.q + uo * x.tc * so;
    gdc[idx] = A;
}

inline hipStream_t ST(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace

extern "C" int dlsg_lstm_cell_fwd(const float* a, int64_t lda, const float* c_prev, float* h, float* c, int rows, int H, void* stream) {
    if (!a || !c_prev || !h || !c || rows < 0 || H < 1 || lda < 4 * (int64_t)H) return DLSG_EINVAL;
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
extern "C" int dlsg_lstm_cell_bwd2(const float* a, int64_t lda, const float* c_prev, const float* dh, const float* dc, const float* u,
                                   const float* uc, float* ga, float* gc_prev, float* gdh, float* gdc, int rows, int H, void* stream) {
    if (!a || !c_prev || !dh || !dc || !u || !uc || !ga || !gc_prev || !gdh || !gdc || rows < 0 || H < 1 || lda < 4 * (int64_t)H)
        return DLSG_EINVAL;
    if (rows == 0) return DLSG_OK;
    hipLaunchKernelGGL(cell_bwd2_kernel, dim3((rows * H + 255) / 256), dim3(256), 0, ST(stream), a, lda, c_prev, dh, dc, u, uc, ga,
                       gc_prev, gdh, gdc, rows, H);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}
