// Persistent BiLSTM recurrence of EncoderVisual (models/layer.py:26,52: nn.LSTM(H, H, bidirectional=True, batch_first=True)).
//
// The input half of the gates (x W_ih^T for all T steps) stays one GEMM (engine.encvis_fwd); this file is the part that is
// sequential in time: for t = 0..T-1, per direction,  gates_t = xg_t + b_ih + b_hh + h_{t-1} W_hh^T ; cell ; h_t.
// As separate launches a step is a K-split skinny GEMM (64 x 4096 x 1024 per direction, every workgroup re-reading its
// weight slice from L2) + a pointwise launch + two kernel boundaries: ~27 us, 25 times per direction.  Here the whole
// sequence of BOTH directions is ONE launch of 2 * H/8 workgroups, one per CU:
//
//   * a workgroup owns 8 hidden units of one direction = 32 gate columns (i, f, g, o of those units).  Its 32 x H slice of
//     W_hh is loaded into LDS ONCE (128 KB at H = 1024) in MFMA-operand order and stays there for all T steps;
//   * the cell state of its 64 x 8 cells never leaves registers;
//   * h_t is exchanged through L2 in k-major form hx[dir][t][unit][batch row] (an owner's 8 units x 64 rows = 2 KB
//     contiguous): written with write-through (sc1) 16-byte stores, drained, then ONE lane stores the workgroup's flag for
//     step t (MI355X_MICROARCH.md, inter-workgroup visibility: every payload store sc1 + vmcnt(0) by every storing wave +
//     barrier + sc1 flag; every payload load sc1 to registers -> no cache-invalidating acquire needed);
//   * consumers: wave w of a workgroup contracts the k range [w H/4, (w+1) H/4) -- the units of H/32 producer
//     workgroups -- so it polls exactly those <= 32 flags (one per lane, relaxed agent-scope loads), then streams its
//     64 rows x H/4 slice of h_{t-1} straight into registers as the MFMA A operand (16-byte coalesced sc1 loads: lane (r, kg)
//     reads rows 4r..4r+3 of one k, which become the same row of four 16-row blocks), against B fragments read from LDS with
//     conflict-free ds_read_b128.  v_mfma_f32_16x16x4_f32: exact fp32;
//   * the four K-partials are summed through 24 KB of LDS so that wave w ends with accumulator register w of every block:
//     rows 16q + 4w + (0..3) -- four consecutive batch rows per lane, i.e. one 16-byte hx store per lane.
//
// Every workgroup must be resident at the same time (grid <= 256 = the CU count; > 128 KB of LDS per workgroup keeps it at
// one per CU).  Spins are bounded: a workgroup that waits longer than ~1 s raises *err and goes on with whatever it read, so
// a mis-scheduled launch ends with a flagged, wrong result instead of a hung device.
//
// Backward through time: bilstm_bwd below, same ownership (8 units per workgroup) but two hand-offs per step.
#include <hip/hip_runtime.h>

#include <mutex>

#include "common.hpp"
#include "dlsg.h"

namespace {

using dlsg::dpp_f32;
using dlsg::sigmoidf_;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int BL_THREADS = 256;        // 4 waves, one per SIMD
constexpr int BL_ROWS = 64;            // batch rows per launch (positions of the exchange buffer)
constexpr unsigned BL_SPIN_LIMIT = 1u << 21;
constexpr int SC1 = 16;                // aux bit of the raw buffer builtins: sc1 (write-through store / L1-bypassing load)

__device__ __forceinline__ f32x4 ld_sc1(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, SC1));
}
__device__ __forceinline__ void st_sc1(__amdgpu_buffer_rsrc_t r, int byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, SC1);
}

// wave-level wait: lanes [0, n) each watch one flag word until all of them read `want`.  Returns false on time-out.
__device__ __forceinline__ bool wait_flags(const uint32_t* flags, int n, uint32_t want) {
    const int lane = threadIdx.x & 63;
    for (unsigned spins = 0;; ++spins) {
        uint32_t v = want;
        if (lane < n) v = __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all(v == want)) break;
        if (spins > BL_SPIN_LIMIT) return false;
        __builtin_amdgcn_s_sleep(4);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      // compiler-only: the payload loads stay below the poll
    return true;
}

// LDS image of a 32-column weight slice as the B operand of v_mfma_f32_16x16x4_f32, read with ds_read_b128:
//   slot(cb, jb, kg, c) = ((cb * NJB + jb) * 4 + kg) * 16 + c   (16-byte slots), holding k = 16 jb + 4 kg + (0..3) of column
//   (cb, c).  A ds_read_b128 is served in four groups of 16 lanes whose (kg, c) pairs cover every c exactly once, and
//   slot mod 16 == c: conflict-free without padding.

// ================================================================================================ forward
// J = H / 64: 16-deep k blocks per wave.
template <int J>
__global__ __launch_bounds__(BL_THREADS) void bilstm_fwd_kernel(const dlsg_bilstm_args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int H = 64 * J, NJB = H / 16, WGS = H / 8;
    f32x4* wimg = reinterpret_cast<f32x4*>(smem);                       // 2 * NJB * 64 slots = 32 * H floats
    float* red = smem + 32 * H;                                          // [dst 4][src' 3][8][64]
    const int d = blockIdx.x / WGS, wg = blockIdx.x % WGS, u0 = wg * 8;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int B = a.B, T = a.T;
    const bool cell_lane = c < 8;
    const int unit = u0 + (c & 7);

    // ---- weights: once per launch (2 J slots per thread, every load in flight before the first LDS write)
    {
        const float* W = a.w_hh[d];
        f32x4 tmp[2 * J];
#pragma unroll
        for (int it = 0; it < 2 * J; ++it) {
            const int s = threadIdx.x + it * BL_THREADS;
            const int cc = s & 15, kg = (s >> 4) & 3, jb = (s >> 6) % NJB, cb = (s >> 6) / NJB;
            const int gate = cb * 2 + (cc >> 3);
            tmp[it] = *reinterpret_cast<const f32x4*>(W + (int64_t)(gate * H + u0 + (cc & 7)) * H + 16 * jb + 4 * kg);
        }
#pragma unroll
        for (int it = 0; it < 2 * J; ++it) wimg[threadIdx.x + it * BL_THREADS] = tmp[it];
    }
    float bias[4] = {0.f, 0.f, 0.f, 0.f};
    if (cell_lane) {
#pragma unroll
        for (int g = 0; g < 4; ++g) bias[g] = a.b_ih[d][g * H + unit] + a.b_hh[d][g * H + unit];
    }
    float cst[4] = {0.f, 0.f, 0.f, 0.f};                                 // cell state of rows 16q + 4w + i, unit `unit`
    const __amdgpu_buffer_rsrc_t hx = __builtin_amdgcn_make_buffer_rsrc(a.hx, 0, 2 * T * H * BL_ROWS * 4, 0x00020000);
    const int row0 = 16 * q + 4 * w;
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        const int tt = d == 0 ? t : T - 1 - t;
        const int nxt = d == 0 ? tt + 1 : tt - 1;
        // ---- x-gates of this step for the lane's 4 cells: in flight while the wave waits for h_{t-1}
        float pre[4][4];                                                 // [gate][row i]
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) pre[g][i] = 0.f;
        if (cell_lane) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int b = row0 + i;
                if (b < B) {
                    const float* xp = a.xg[d] + ((int64_t)b * T + tt) * a.ldxg + unit;
#pragma unroll
                    for (int g = 0; g < 4; ++g) pre[g][i] = xp[g * H];
                }
            }
        }
        if (t > 0) {
            // ---- wait for the producers of this wave's k range, then stream h_{t-1}[:, k range] into registers
            const uint32_t* fl = a.flags + (int64_t)(d * T + (t - 1)) * WGS + w * (2 * J);
            if (!wait_flags(fl, 2 * J, 1u) && lane == 0 && a.err) atomicExch(a.err, 1);
            const int abase = (((d * T + (t - 1)) * H + w * (H / 4) + 4 * q) * BL_ROWS + 4 * c) * 4;     // bytes; + (16 j + s) * 256
            f32x4 av[J][4];
#pragma unroll
            for (int j = 0; j < J; ++j)
#pragma unroll
                for (int s = 0; s < 4; ++s) av[j][s] = ld_sc1(hx, abase + (16 * j + s) * BL_ROWS * 4);
            // all 4 J loads (64 KB per wave at H = 1024) are in flight before the first MFMA: the scheduler must not sink them
            // next to their uses (it otherwise emits load -> vmcnt(0) -> 8 MFMAs, one L2 round trip per 16-byte piece)
            __builtin_amdgcn_sched_barrier(0);
            f32x4 acc[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) { acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = acc[i][0]; }
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int jb = w * J + j;
                const f32x4 b0 = wimg[(jb * 4 + q) * 16 + c];
                const f32x4 b1 = wimg[((NJB + jb) * 4 + q) * 16 + c];
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][s][i], b0[s], acc[i][0], 0, 0, 0);
                        acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][s][i], b1[s], acc[i][1], 0, 0, 0);
                    }
            }
            // ---- sum the four K-partials: wave `dst` ends up owning accumulator register `dst` of every block
#pragma unroll
            for (int dst = 0; dst < 4; ++dst) {
                if (dst != w) {
                    float* p = red + ((dst * 3 + (w - (w > dst))) * 8) * 64 + lane;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { p[(2 * i) * 64] = acc[i][0][dst]; p[(2 * i + 1) * 64] = acc[i][1][dst]; }
                }
            }
            __syncthreads();
            float own[4][2];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r == w) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { own[i][0] = acc[i][0][r]; own[i][1] = acc[i][1][r]; }
                }
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) {
                const float* p = red + ((w * 3 + s3) * 8) * 64 + lane;
#pragma unroll
                for (int i = 0; i < 4; ++i) { own[i][0] += p[(2 * i) * 64]; own[i][1] += p[(2 * i + 1) * 64]; }
            }
            // lane c < 8 holds gates i (cb 0) and g (cb 1) of its unit; f and o sit 8 lanes up in the same row of 16
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float f_ = dpp_f32<0x128>(own[i][0]);              // row_ror:8
                const float o_ = dpp_f32<0x128>(own[i][1]);
                pre[0][i] += own[i][0]; pre[1][i] += f_; pre[2][i] += own[i][1]; pre[3][i] += o_;
            }
        }
        // ---- cell (only lanes c < 8 hold real cells)
        f32x4 hv;
        float gi[4], gf[4], gg[4], go[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            gi[i] = sigmoidf_(pre[0][i] + bias[0]);
            gf[i] = sigmoidf_(pre[1][i] + bias[1]);
            gg[i] = tanhf(pre[2][i] + bias[2]);
            go[i] = sigmoidf_(pre[3][i] + bias[3]);
            cst[i] = gf[i] * cst[i] + gi[i] * gg[i];
            hv[i] = go[i] * tanhf(cst[i]);
        }
        // ---- publish h_t: write-through stores, drained by every wave, then one flag
        if (cell_lane) st_sc1(hx, (((d * T + t) * H + unit) * BL_ROWS + row0) * 4, hv);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_store(a.flags + (int64_t)(d * T + t) * WGS + wg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ---- the rest goes out with plain stores while the next step is already waiting
        if (cell_lane) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int b = row0 + i;
                if (b < B) {
                    const int64_t bt = (int64_t)b * T + tt;
                    a.out[bt * (2 * H) + d * H + unit] = hv[i];
                    a.c[d][bt * H + unit] = cst[i];
                    if (nxt >= 0 && nxt < T) a.hprev[d][((int64_t)b * T + nxt) * H + unit] = hv[i];
                    float* gp = a.gates[d] + bt * (4 * H) + unit;
                    gp[0] = gi[i]; gp[H] = gf[i]; gp[2 * H] = gg[i]; gp[3 * H] = go[i];
                }
            }
        }
    }
}

template <int J>
int launch_fwd(const dlsg_bilstm_args* a, hipStream_t st) {
    constexpr int H = 64 * J;
    constexpr int lds_bytes = (32 * H + 4 * 3 * 8 * 64) * 4;
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bilstm_fwd_kernel<J>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    });
    hipLaunchKernelGGL((bilstm_fwd_kernel<J>), dim3(2 * (H / 8)), dim3(BL_THREADS), lds_bytes, st, *a);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}

// ================================================================================================ backward through time
// dG_t = cell'(gates_t, c_t, c_{t-1}; dh_t, dc_t) ;  dh_{t-1} += dG_t W_hh  (M = 64 rows, N = H, K = 4H per direction and step).
// Workgroup wg = 4 n + kq of a direction (H/8 of them):
//   * pointwise owner of the 8 units 8 wg .. 8 wg + 7 (all four gates; dc stays in registers);
//   * product owner of output units 32 n .. 32 n + 31 over the K quarter of gate kq: its 32 x H slice of W_hh^T
//     (W_hh[kq H + k'][32 n + j]) is resident in LDS, so an MFMA tile is 64 x 32 with both column blocks full (an 8-unit
//     owner contracting all 4H columns would waste half of every MFMA and pull 1 MB per step).
// Two hand-offs per step, both through per-step buffers with write-through stores + flags (see the forward):
//   (a) gx[dir][s][gate][unit][row]: every workgroup publishes dG of its 8 units (8 KB); wave w of a consumer waits for the
//       H/32 producers of its k' range of gate kq and streams 64 KB of it as the MFMA A operand;
//   (b) px[dir][s][kq][unit][row]: every workgroup publishes its 64 x 32 partial of dh_{t-1} (8 KB); the pointwise owner of a
//       unit sums the four K-quarter partials of the four workgroups (n, 0..3).
template <int J>
__global__ __launch_bounds__(BL_THREADS) void bilstm_bwd_kernel(const dlsg_bilstm_bwd_args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int H = 64 * J, NJB = H / 16, WGS = H / 8;
    f32x4* wimg = reinterpret_cast<f32x4*>(smem);
    float* red = smem + 32 * H;
    const int d = blockIdx.x / WGS, wg = blockIdx.x % WGS, n = wg >> 2, kq = wg & 3;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int B = a.B, T = a.T;
    // pointwise cells of this thread: unit 8 wg + pj, rows 2 pp, 2 pp + 1
    const int pj = threadIdx.x >> 5, pp = threadIdx.x & 31, punit = 8 * wg + pj;

    // ---- W_hh^T slice: slot (cb, jb, kg, cc) <- W_hh[kq H + 16 jb + 4 kg + e][32 n + 16 cb + cc], e = 0..3
    {
        const float* W = a.w_hh[d] + (int64_t)kq * H * H + 32 * n;
#pragma unroll 4
        for (int it = 0; it < 2 * J; ++it) {
            const int s = threadIdx.x + it * BL_THREADS;
            const int cc = s & 15, kg = (s >> 4) & 3, jb = (s >> 6) % NJB, cb = (s >> 6) / NJB;
            const float* src = W + (int64_t)(16 * jb + 4 * kg) * H + 16 * cb + cc;
            wimg[s] = f32x4{src[0], src[H], src[2 * H], src[3 * H]};
        }
    }
    const int xbytes = 2 * T * 4 * H * BL_ROWS * 4;
    const __amdgpu_buffer_rsrc_t gx = __builtin_amdgcn_make_buffer_rsrc(a.gx, 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t px = __builtin_amdgcn_make_buffer_rsrc(a.px, 0, xbytes, 0x00020000);
    const uint32_t* fa = a.flags;                                        // [d][s][wg]: dG of step s published
    const uint32_t* fb = a.flags + 2 * T * WGS;                          // [d][s][wg]: partial dh of step s published
    float dc[2] = {0.f, 0.f};
    __syncthreads();

    for (int s = 0; s < T; ++s) {
        const int fstep = T - 1 - s;                                     // forward step index being differentiated
        const int t = d == 0 ? fstep : T - 1 - fstep;
        const int tp = d == 0 ? t - 1 : t + 1;                           // time index of the previous forward step
        // ---- saved activations and the incoming gradient of this thread's two cells: in flight during the flag wait
        float gi[2], gf[2], gg[2], go[2], cc_[2], cp[2], dh[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int b = 2 * pp + i;
            gi[i] = gf[i] = gg[i] = go[i] = cc_[i] = cp[i] = dh[i] = 0.f;
            if (b < B) {
                const int64_t bt = (int64_t)b * T + t;
                const float* gp = a.gates[d] + bt * (4 * H) + punit;
                gi[i] = gp[0]; gf[i] = gp[H]; gg[i] = gp[2 * H]; go[i] = gp[3 * H];
                cc_[i] = a.c[d][bt * H + punit];
                if (fstep > 0) cp[i] = a.c[d][((int64_t)b * T + tp) * H + punit];
                dh[i] = a.dout[bt * (2 * H) + d * H + punit];
            }
        }
        if (s > 0) {
            // (b) the four K-quarter partials of dh for this thread's unit
            const uint32_t* fl = fb + (int64_t)(d * T + (s - 1)) * WGS + 4 * n;
            if (!wait_flags(fl, 4, 1u) && lane == 0 && a.err) atomicExch(a.err, 2);
            const int pbase = ((((d * T + (s - 1)) * 4) * H + punit) * BL_ROWS + 2 * pp) * 4;
            f32x2 part[4];
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4)
                part[k4] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(px, pbase + k4 * H * BL_ROWS * 4, 0, SC1));
            dh[0] += (part[0].x + part[1].x) + (part[2].x + part[3].x);
            dh[1] += (part[0].y + part[1].y) + (part[2].y + part[3].y);
        }
        // ---- cell backward
        float dg[4][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float tc = tanhf(cc_[i]);
            const float dct = dh[i] * go[i] * (1.f - tc * tc) + dc[i];
            dg[0][i] = dct * gg[i] * gi[i] * (1.f - gi[i]);
            dg[1][i] = dct * cp[i] * gf[i] * (1.f - gf[i]);
            dg[2][i] = dct * gi[i] * (1.f - gg[i] * gg[i]);
            dg[3][i] = dh[i] * tc * go[i] * (1.f - go[i]);
            dc[i] = dct * gf[i];
        }
        if (fstep > 0) {
            // (a) publish dG of this thread's cells, k-major, then the workgroup's flag
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x2 v = f32x2{dg[g][0], dg[g][1]};
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), gx,
                                                      ((((d * T + s) * 4 + g) * H + punit) * BL_ROWS + 2 * pp) * 4, 0, SC1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0)
                __hip_atomic_store(const_cast<uint32_t*>(fa) + (int64_t)(d * T + s) * WGS + wg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // row-major dG for the weight-gradient products after the loop: plain stores, nobody in this launch reads them
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int b = 2 * pp + i;
            if (b < B) {
                float* gp = a.dgates[d] + ((int64_t)b * T + t) * (4 * H) + punit;
                gp[0] = dg[0][i]; gp[H] = dg[1][i]; gp[2 * H] = dg[2][i]; gp[3 * H] = dg[3][i];
            }
        }
        if (fstep == 0) break;
        // ---- partial dh_{t-1}[:, 32 n .. 32 n + 31] over gate kq: wave w contracts k' in [w H/4, (w+1) H/4)
        {
            const uint32_t* fl = fa + (int64_t)(d * T + s) * WGS + w * (2 * J);
            if (!wait_flags(fl, 2 * J, 1u) && lane == 0 && a.err) atomicExch(a.err, 3);
            const int abase = ((((d * T + s) * 4 + kq) * H + w * (H / 4) + 4 * q) * BL_ROWS + 4 * c) * 4;
            f32x4 av[J][4];
#pragma unroll
            for (int j = 0; j < J; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) av[j][e] = ld_sc1(gx, abase + (16 * j + e) * BL_ROWS * 4);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 acc[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) { acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = acc[i][0]; }
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int jb = w * J + j;
                const f32x4 b0 = wimg[(jb * 4 + q) * 16 + c];
                const f32x4 b1 = wimg[((NJB + jb) * 4 + q) * 16 + c];
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][e][i], b0[e], acc[i][0], 0, 0, 0);
                        acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][e][i], b1[e], acc[i][1], 0, 0, 0);
                    }
            }
#pragma unroll
            for (int dst = 0; dst < 4; ++dst) {
                if (dst != w) {
                    float* p = red + ((dst * 3 + (w - (w > dst))) * 8) * 64 + lane;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { p[(2 * i) * 64] = acc[i][0][dst]; p[(2 * i + 1) * 64] = acc[i][1][dst]; }
                }
            }
            __syncthreads();
            float own[4][2];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r == w) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { own[i][0] = acc[i][0][r]; own[i][1] = acc[i][1][r]; }
                }
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) {
                const float* p = red + ((w * 3 + s3) * 8) * 64 + lane;
#pragma unroll
                for (int i = 0; i < 4; ++i) { own[i][0] += p[(2 * i) * 64]; own[i][1] += p[(2 * i + 1) * 64]; }
            }
            // lane (c, q) of wave w: rows 16 q + 4 w + (0..3) of output units 32 n + c and 32 n + 16 + c
            const int obase = ((((d * T + s) * 4 + kq) * H + 32 * n + c) * BL_ROWS + 16 * q + 4 * w) * 4;
            st_sc1(px, obase, f32x4{own[0][0], own[1][0], own[2][0], own[3][0]});
            st_sc1(px, obase + 16 * BL_ROWS * 4, f32x4{own[0][1], own[1][1], own[2][1], own[3][1]});
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0)
                __hip_atomic_store(const_cast<uint32_t*>(fb) + (int64_t)(d * T + s) * WGS + wg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int J>
int launch_bwd(const dlsg_bilstm_bwd_args* a, hipStream_t st) {
    constexpr int H = 64 * J;
    constexpr int lds_bytes = (32 * H + 4 * 3 * 8 * 64) * 4;
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bilstm_bwd_kernel<J>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    });
    hipLaunchKernelGGL((bilstm_bwd_kernel<J>), dim3(2 * (H / 8)), dim3(BL_THREADS), lds_bytes, st, *a);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}

}  // namespace

// Every workgroup of the launch must be resident at once (they wait for each other inside the kernel): the current device
// needs at least 2 * H / 8 compute units (a partitioned or CU-masked device with fewer gets the per-step schedule instead).
static int device_cus() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return cus;
}
extern "C" int dlsg_bilstm_supported(int B, int T, int H) {
    if (!(B >= 1 && B <= BL_ROWS && T >= 1 && T <= 4096 && (H == 64 || H == 512 || H == 1024))) return 0;
    // buffer sizes and byte offsets are 32-bit: the backward's exchange buffers are the largest (2 T x 4H x 64 rows x 4 bytes)
    if ((int64_t)2 * T * 4 * H * BL_ROWS * 4 >= (int64_t)1 << 31) return 0;
    return device_cus() >= 2 * (H / 8) ? 1 : 0;
}
extern "C" int64_t dlsg_bilstm_hx_floats(int T, int H) { return (int64_t)2 * T * H * BL_ROWS; }
extern "C" int64_t dlsg_bilstm_flag_words(int T, int H) { return ((int64_t)2 * T * (H / 8) + 3) / 4 * 4; }

extern "C" int dlsg_bilstm_fwd(const dlsg_bilstm_args* a, void* stream) {
    if (!a || !dlsg_bilstm_supported(a->B, a->T, a->H) || !a->hx || !a->flags || a->ldxg < 4 * a->H) return DLSG_EINVAL;
    for (int d = 0; d < 2; ++d)
        if (!a->xg[d] || !a->w_hh[d] || !a->b_ih[d] || !a->b_hh[d] || !a->hprev[d] || !a->c[d] || !a->gates[d]) return DLSG_EINVAL;
    if ((reinterpret_cast<uintptr_t>(a->hx) & 15) || (reinterpret_cast<uintptr_t>(a->w_hh[0]) & 15) ||
        (reinterpret_cast<uintptr_t>(a->w_hh[1]) & 15))
        return DLSG_EALIGN;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // the flags must read 0 before the first workgroup polls them: a memset node in front of the launch (replayed with it)
    if (hipMemsetAsync(a->flags, 0, dlsg_bilstm_flag_words(a->T, a->H) * 4, st) != hipSuccess) return DLSG_ELAUNCH;
    switch (a->H) {
        case 64: return launch_fwd<1>(a, st);
        case 512: return launch_fwd<8>(a, st);
        case 1024: return launch_fwd<16>(a, st);
    }
    return DLSG_EINVAL;
}

extern "C" int64_t dlsg_bilstm_bwd_x_floats(int T, int H) { return (int64_t)2 * T * 4 * H * BL_ROWS; }

extern "C" int dlsg_bilstm_bwd(const dlsg_bilstm_bwd_args* a, void* stream) {
    if (!a || !dlsg_bilstm_supported(a->B, a->T, a->H) || !a->gx || !a->px || !a->flags || !a->dout) return DLSG_EINVAL;
    for (int d = 0; d < 2; ++d)
        if (!a->gates[d] || !a->c[d] || !a->w_hh[d] || !a->dgates[d]) return DLSG_EINVAL;
    if ((reinterpret_cast<uintptr_t>(a->gx) & 15) || (reinterpret_cast<uintptr_t>(a->px) & 15)) return DLSG_EALIGN;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (hipMemsetAsync(a->flags, 0, 2 * dlsg_bilstm_flag_words(a->T, a->H) * 4, st) != hipSuccess) return DLSG_ELAUNCH;
    switch (a->H) {
        case 64: return launch_bwd<1>(a, st);
        case 512: return launch_bwd<8>(a, st);
        case 1024: return launch_bwd<16>(a, st);
    }
    return DLSG_EINVAL;
}
