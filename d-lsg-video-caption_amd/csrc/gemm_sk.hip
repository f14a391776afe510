// Persistent stream-K fp32 GEMM: ONE launch of (at most) one workgroup per CU for the products that fill the chip --
// the region projections (layer.py:184), their weight gradients, the mid-size weight-gradient groups.  256 x 256 output
// tile on v_mfma_f32_32x32x2_f32 (exact fp32; per output element the same k-ordered fmaf chain as gemm.hip / gemm_big.hip,
// except on the tiles whose contraction is split between workgroups, where the partial chains are added in a fixed order).
//
// Schedule (host: sk_plan).  The launch has T tiles (all groups) and P workgroups, T = R * P + rem:
//   * the last `rem` tiles are "split tiles": their stages are dealt to the workgroups in equal consecutive runs (a run lies in
//     one tile or ends one and begins the next), and every workgroup STARTS with its run, so all of them stay busy for the
//     same time.  A share is written to a workspace slot in accumulator order (16-B stores) and published (release + one
//     counter per split tile).  It is finished at the END of the kernel: contributor c of n adds up the n slots of every
//     sub-block k = c (mod n) (16 sub-blocks of 32 x 32 per wave) in contributor order and runs the epilogue on them.  R whole
//     tiles lie between the share's store and that point, so the tile is complete by then -- and if it is not (the launch is
//     not co-resident: another tenant on the device), the workgroup does not wait: it leaves its sub-blocks to the contributor
//     that finishes last.  No workgroup ever waits for another one: such a launch is slower, never stuck;
//   * then R rounds of whole tiles, workgroup v (virtual id: the 32 workgroups an XCD receives are consecutive) takes tile
//     j * P + v: the workgroups of one XCD walk one compact block of row panels x column tiles in lock step, so the XCD's
//     L2 serves every staged A / B block to several of them.
// Main loop (per 32-deep stage: 256 MFMAs per wave; LDS: two 64-KB stage buffers filled by LDS-DMA):
//   q0 | q1 | q2 | wait + barrier | q3      (q = 8 k = 64 MFMAs)
// the barrier sits BETWEEN q2 and q3: at it, this stage's buffer has been read out completely (q3's fragments are in
// registers) and the next stage's buffer has landed; q3 then reads the next stage's first fragments under its own MFMAs and
// the DMA of the stage after next starts into the buffer that just became free -- the matrix pipe never waits for an LDS
// round trip, and a stage's bytes are requested 1.3 stages (~9 us) before they are used.  Fragment reads and DMA pieces are
// pinned between groups of four MFMAs (sched_barrier): left to the scheduler they sit in a clump in front of each q block
// behind an lgkmcnt(0) (gemm_big.hip: 5-10 % of the loop).  Tile changes happen inside the same stage stream: the next
// tile's first stages are in flight while the finished tile is stored, and its stores drain under the next tile's MFMAs.
#include <mutex>
#include <type_traits>

#include "common.hpp"
#include "dlsg.h"

namespace {

constexpr int SK_THREADS = 256;
constexpr int SK_BN = 256, SK_BK = 32;             // tile: BM x BN, (BM, BN) = (256, 256), (128, 256) or (128, 128) (template parameters);
                                                   // SK_BN = the widest; 32-deep stages
constexpr int SK_SLOT_BYTES = 256 * SK_BN * 4;     // a workspace slot holds the accumulators of the larger tile
constexpr int SK_CNT_BYTES = 16384;                // in front of the slots: counters (2 words per split tile) in the first 4 KB; the
                                                   // rest is where a PROBES=1 build leaves its timestamps
constexpr int SK_MAXSPLIT = 16;                    // contributors per split tile (= sub-blocks a wave can hand out)
constexpr int SK_PATIENCE = 64;                    // polls (s_sleep(8) + one L2 round trip each: ~15 us in all) a contributor spends on an
                                                   // incomplete split tile before it leaves its sub-blocks to the tile's last finisher

struct SkGroup {
    const float* A; const float* B; float* C; const float* bias;
    int64_t lda, ldb, ldc;
    int32_t nst, tiles_n, N, tile0;                // stages (K / 32), column tiles, output width, first linear tile id
};
struct SkArgs {
    int32_t M, ngroups, flags, P;
    int32_t T, rounds, rem, sk_wgs;                // T = rounds * P + rem; sk_wgs: workgroups that share the last rem tiles
    int32_t sk_nst, bm;                            // stages per tile of those tiles when they are split (0: one whole tile each); tile height
    int32_t bn, pad3_;                             // tile width
    float alpha; int32_t xmap;                     // > 1: the launch has no whole-tile rounds and every tile exactly xmap contributors -- see the kernel's virtual id
    const int32_t* skip_if;
    float* slots; uint32_t* cnt; int32_t* err;
    SkGroup g[DLSG_GEMM_MAXG];
};

struct Item {                                      // one contiguous stage range of one tile; every field wave-uniform
    int32_t gi, m0, n0, Ng;
    int32_t s0, s1, kind, r;                       // kind: 1 whole tile, 2 share of split tile r (its slot: 2 v + item index)
};

typedef __attribute__((address_space(3))) void* sk_lp_t;
typedef const __attribute__((address_space(1))) void* sk_gp_t;

// One LDS-DMA piece: 64 lanes x 16 B from (uniform base + per-lane 32-bit offset) to LDS at `lds_addr` + 16 * lane.  Written as
// inline assembly on purpose: as the builtin, the compiler orders every later ds_read behind it with an s_waitcnt vmcnt(0)
// (it cannot tell the stage buffer being filled from the one being read) -- one in the first fragment reads of every stage,
// 0.5 us after the pieces were issued.  Its own vmcnt bookkeeping does not see these requests; the waits it places for its
// own loads can only become stricter by that (the counter is in-order), and the stage loop waits for the pieces itself.
__device__ __forceinline__ void sk_glds16(const char* base, uint32_t voff, uint32_t lds_addr) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds_addr), "v"(voff), "s"(base) : "memory", "m0");
#endif
}
__device__ __forceinline__ int sk_u(int x) { return __builtin_amdgcn_readfirstlane(x); }       // pin a uniform value to an SGPR
__device__ __forceinline__ const char* sk_up(const void* q) {
    const uint64_t u = reinterpret_cast<uint64_t>(q);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}

#ifdef DLSG_PROBES
// diagnostic build (make PROBES=1): workgroups v < 8 leave 100-MHz timestamps in the upper half of the counter area
#define SK_STAMP(i) do { if (v < 8 && threadIdx.x == 0 && (i) < 32) reinterpret_cast<uint64_t*>(p.cnt + 1024)[v * 32 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
// ... and workgroup 0 one per stage (the first 1024 stages)
#define SK_STAGE_STAMP(i) do { if (v == 0 && threadIdx.x == 0 && (i) < 1024) reinterpret_cast<uint64_t*>(p.cnt + 2048)[(i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SK_STAMP(i) do { } while (0)
#define SK_STAGE_STAMP(i) do { } while (0)
#endif

// The split tiles' stages, U = rem * sk_nst of them, are dealt to the sk_wgs workgroups in equal consecutive runs: workgroup v
// has [v U / sk_wgs, (v + 1) U / sk_wgs) -- a run is shorter than a tile, so it lies in one tile or ends one and begins the next.
__device__ __forceinline__ int64_t sk_run0(const SkArgs& p, int v) { return (int64_t)v * p.rem * p.sk_nst / p.sk_wgs; }
// items of the split phase of workgroup v: 0 (not taking part), 1 or 2
__device__ __forceinline__ int sk_nsplit(const SkArgs& p, int v) {
    if (v >= p.sk_wgs) return 0;
    if (p.sk_nst == 0) return 1;
    const int64_t u0 = sk_run0(p, v), u1 = sk_run0(p, v + 1);
    return (u1 - 1) / p.sk_nst != u0 / p.sk_nst ? 2 : 1;
}
// contributors of split tile r: workgroups [lo, hi]
__device__ __forceinline__ void sk_contributors(const SkArgs& p, int r, int& lo, int& hi) {
    const int64_t U = (int64_t)p.rem * p.sk_nst, b = (int64_t)r * p.sk_nst, e = b + p.sk_nst;
    lo = (int)(b * p.sk_wgs / U);
    while (lo > 0 && sk_run0(p, lo) > b) --lo;
    while (sk_run0(p, lo + 1) <= b) ++lo;
    hi = (int)((e - 1) * p.sk_wgs / U);
    while (hi + 1 < p.sk_wgs && sk_run0(p, hi + 1) < e) ++hi;
    while (sk_run0(p, hi) >= e) --hi;
}

// item idx of virtual workgroup v (idx < its item count): its one or two shares of split tiles first, then one whole tile per round
__device__ __forceinline__ Item sk_item(const SkArgs& p, int v, int idx) {
    Item it;
    it.r = 0; it.kind = 1;
    const int ns = sk_nsplit(p, v);
    int tile, s0 = 0, s1 = -1;
    if (idx < ns) {
        if (p.sk_nst == 0) it.r = v;
        else {
            const int64_t u0 = sk_run0(p, v), u1 = sk_run0(p, v + 1);
            const int r0 = (int)(u0 / p.sk_nst);
            it.r = r0 + idx;
            s0 = idx == 0 ? (int)(u0 - (int64_t)r0 * p.sk_nst) : 0;
            s1 = (idx == 0 && ns == 2) ? p.sk_nst : (int)(u1 - (int64_t)it.r * p.sk_nst);
            it.kind = (s0 == 0 && s1 == p.sk_nst) ? 1 : 2;
        }
        tile = p.rounds * p.P + it.r;
    } else {
        tile = (idx - ns) * p.P + v;
    }
    int gi = 0;
    for (int i = 1; i < p.ngroups; ++i) gi = (tile >= p.g[i].tile0) ? i : gi;
    const SkGroup& g = p.g[gi];
    const int local = tile - g.tile0;
    const int tm = local / g.tiles_n, tn = local - tm * g.tiles_n;
    it.gi = sk_u(gi); it.m0 = sk_u(tm * p.bm); it.n0 = sk_u(tn * p.bn); it.Ng = sk_u(g.N);
    if (s1 < 0) { s0 = 0; s1 = g.nst; }
    it.s0 = sk_u(s0); it.s1 = sk_u(s1); it.kind = sk_u(it.kind); it.r = sk_u(it.r);
    return it;
}

// per-lane byte offsets of this wave's ROWS / 32 DMA pieces (1 KB each) of one operand, relative to (tile base + stage offset)
//   T == false: element (row, k) at base[row * ld + k] -> image [row][32], 16-B segments swizzled: slot (row, s) holds
//               k-segment s ^ ((row >> 1) & 7); piece = 8 rows
//   T == true : element (row, k) at base[k * ld + row] -> image [k][ROWS]; piece = 256 / ROWS k rows
template <bool T, int ROWS>
__device__ __forceinline__ void sk_voff(int (&voff)[ROWS / 32], int w, int lane, int row0, int rmax, int64_t ld) {
    constexpr int PW = ROWS / 32;                  // pieces per wave
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int pc = w * PW + i;
        if (!T) {
            const int R = 8 * pc + (lane >> 3);
            const int seg = (lane & 7) ^ ((R >> 1) & 7);
            voff[i] = (int)((min(row0 + R, rmax - 1) - row0) * ld * 4) + 16 * seg;
        } else {
            constexpr int KR = 256 / ROWS, LPR = 64 / KR;
            const int kr = KR * pc + lane / LPR;
            voff[i] = (int)(kr * ld * 4) + (min(row0 + 4 * (lane % LPR), rmax - 4) - row0) * 4;
        }
    }
}

constexpr int SK_BOUNCE = 32 * 32 * 4;           // bytes of a wave's bounce tile in LDS

// tanh x = 1 - 2 / (1 + e^{2x}) on the hardware's exp2 / rcp (1 ulp each): absolute error ~1e-7, saturates to +-1 through
// e^{2x} -> inf / 0.  tanhf (~60 instructions) cost 25 us per 256 x 256 tile, a third of the tile's epilogue.
__device__ __forceinline__ float sk_tanh(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + e);
}

// Epilogue of one 32 x 32 sub-block (i, j) of a wave.  `val` is in the C/D layout of the 32x32 MFMA (col = lane & 31, row =
// e' + 4 (2 e4 + (lane >> 5)) for element e = 4 e4 + e'): stored that way, a wave instruction writes 4 bytes per lane into two
// rows -- 42 us per 256-KB tile with every CU storing at once, against 12 us for the same bytes in 16-B pieces.  So the
// sub-block goes through the wave's own 4-KB bounce tile in LDS: a lane writes its four row quads as 16-B pieces (straight
// from the accumulator file: four consecutive accumulators are one ds_write_b128; any arithmetic on them first makes the
// register allocator copy all 256 accumulators out at the top of the block), tile layout [row quad G][column][row in quad];
// lane (G, cs) = (lane >> 3, lane & 7) reads the 4 x 4 block of row quad G and columns 4 cs .. 4 cs + 3 back as 64 contiguous
// bytes and stores its four rows as 16-B pieces: 8 rows x 128 B per store instruction.  One wave's DS operations execute in
// order: no barrier.  bias4: the lane's bias values of its four columns, fetched by the caller before the tile's first store.
template <class V>
__device__ __forceinline__ void sk_store_block(const SkArgs& p, const SkGroup& g, const Item& it, const V& val, int i, int j,
                                               int w, int lane, f32x4 bias4, float* bounce) {
    const int r = lane & 31, h = lane >> 5, wm = w >> 1, wn = w & 1;
    const int WM = p.bm >> 1;                                 // rows of a wave's quadrant
    f32x4* bw = reinterpret_cast<f32x4*>(bounce) + h * 32 + r;
#pragma unroll
    for (int e4 = 0; e4 < 4; ++e4) {
        f32x4 t;
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = val[4 * e4 + e];
        bw[e4 * 64] = t;
    }
    const int G = lane >> 3, cs = lane & 7;
    const f32x4* br = reinterpret_cast<const f32x4*>(bounce) + G * 32 + 4 * cs;
    f32x4 x[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) x[c] = br[c];
    const int col = it.n0 + wn * (p.bn >> 1) + j * 32 + 4 * cs;
    const int row0 = it.m0 + wm * WM + i * 32 + 4 * G;
    const bool accum = p.flags & DLSG_GEMM_ACCUM, do_tanh = p.flags & DLSG_GEMM_TANH;
    const bool col_ok = col < it.Ng;                          // (widths are multiples of 4: the whole piece is in or out)
    float* cp = g.C + (int64_t)row0 * g.ldc + col;
    // C += (rare: the engine accumulates only into gradients that already hold a contribution): every loaded value is consumed
    // below whether or not its row is stored, so no request is left pending when the stage loop is entered again
    f32x4 old[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        old[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (accum && col_ok && row0 + q < p.M) old[q] = *reinterpret_cast<const f32x4*>(cp + (int64_t)q * g.ldc);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 y = f32x4{x[0][q], x[1][q], x[2][q], x[3][q]};
        y = p.alpha * y + bias4 + old[q];
        if (do_tanh) {
#pragma unroll
            for (int c = 0; c < 4; ++c) y[c] = sk_tanh(y[c]);
        }
        if (col_ok && row0 + q < p.M) *reinterpret_cast<f32x4*>(cp + (int64_t)q * g.ldc) = y;
    }
}

// this lane's bias values for the four column blocks j of its wave (zeros without a bias or past the group's width)
template <int TNB>
__device__ __forceinline__ void sk_bias(const SkArgs& p, const SkGroup& g, const Item& it, int w, int lane, f32x4 (&bias4)[TNB]) {
    const bool use_bias = (p.flags & DLSG_GEMM_BIAS) && g.bias != nullptr;
#pragma unroll
    for (int j = 0; j < TNB; ++j) {
        const int col = it.n0 + (w & 1) * (p.bn >> 1) + j * 32 + 4 * (lane & 7);
        bias4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (use_bias && col < it.Ng) bias4[j] = *reinterpret_cast<const f32x4*>(g.bias + col);
    }
}

template <int BM, int BN, bool AT, bool BT>
__global__ __launch_bounds__(SK_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_sk_kernel(const SkArgs p) {
    constexpr int TM = BM / 64, TNB = BN / 64;     // 32-row / 32-column blocks of a wave (its quadrant is BM / 2 x BN / 2)
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int SK_BBYTES = BN * 128;
    constexpr int SK_ABYTES = BM * 128, SK_STAGE = SK_ABYTES + SK_BBYTES;
    constexpr int PA = BM / 32, PB = BN / 32, PT = PA + PB;              // DMA pieces per wave and stage
    constexpr int NSUB = TNB * TM;                 // 32 x 32 sub-blocks of a wave
    extern __shared__ __attribute__((aligned(16))) char sk_lds[];           // 2 stage buffers
    if (p.skip_if && *p.skip_if) return;                                    // block-uniform
    const int lane = threadIdx.x & 63;
    const int w = sk_u(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wm = w >> 1, wn = w & 1;
    // virtual id: the workgroups an XCD receives (blockIdx % 8) are consecutive -- speed only.  A launch without whole-tile rounds
    // whose tiles all have n = xmap contributors (the deep weight gradients: 64 tiles x 4 K-quarters): an XCD takes ONE K-slice
    // of a compact block of tiles instead of all slices of eight tiles -- its L2 then holds a quarter of the contraction range of
    // both operands (82 MB per XCD instead of the whole of B: 218 MB), the contributors of a tile sit on n different XCDs (their
    // shares travel through memory either way)
    const int bid = blockIdx.x;
    int v_ = (p.P & 7) ? bid : (bid & 7) * (p.P >> 3) + (bid >> 3);
    if (p.xmap > 1) v_ = (((bid & 7) / p.xmap) * (p.P >> 3) + (bid >> 3)) * p.xmap + ((bid & 7) % p.xmap);
    const int v = sk_u(v_);
    const int n_items = sk_u(p.rounds + sk_nsplit(p, v));
    if (n_items == 0) return;
    SK_STAMP(0);
    const uint32_t lds0 = sk_u((uint32_t)reinterpret_cast<uintptr_t>((sk_lp_t)sk_lds));      // LDS byte address of the stage buffers

    f32x16 acc[TM][TNB];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TNB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment byte offsets of k-group q inside a stage buffer (lane (r, h) feeds k = 4 (2q + h) + j to MFMA j of group q)
    int offA[4], offB[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ks = (2 * q + h) ^ ((r >> 1) & 7);
        offA[q] = AT ? ((4 * (2 * q + h)) * BM + wm * WM + r) * 4 : (wm * WM + r) * 128 + ks * 16;
        offB[q] = SK_ABYTES + (BT ? ((4 * (2 * q + h)) * BN + wn * WN + r) * 4 : (wn * WN + r) * 128 + ks * 16);
    }
    f32x4 fa[2][TM], fb[2][TNB];
    auto read_frag = [&](int buf, const char* st, int q, int f) {          // f < TM: A fragment f, else B fragment f - TM
        if (f < TM) {
            if (!AT) fa[buf][f] = *reinterpret_cast<const f32x4*>(st + offA[q] + f * 32 * 128);
            else {
#pragma unroll
                for (int j = 0; j < 4; ++j) fa[buf][f][j] = *reinterpret_cast<const float*>(st + offA[q] + (j * BM + f * 32) * 4);
            }
        } else {
            const int g = f - TM;
            if (!BT) fb[buf][g] = *reinterpret_cast<const f32x4*>(st + offB[q] + g * 32 * 128);
            else {
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[buf][g][j] = *reinterpret_cast<const float*>(st + offB[q] + (j * BN + g * 32) * 4);
            }
        }
    };

    // ---- load cursor: the stage whose DMA is issued next.  Past the last stage of the stream it stays on that stage: the
    // pieces it keeps issuing land in buffers nobody reads any more (no branches in the stage body)
    int li = 0, l_left = 0, lpar = 0;              // item index, stages of the item not yet issued, buffer parity
    int voffA[PA], voffB[PB];
    const char* lA = nullptr; const char* lB = nullptr;                     // stage bases
    int64_t sA = 0, sB = 0;                                                  // bytes per stage
    auto l_enter = [&](int idx) {
        // (rare path: its inputs go through empty asm statements so that nothing computed from them is hoisted out of the stage
        //  loop and kept in registers across it -- with 256 accumulators the loop has none to spare)
        int lane_ = lane, w_ = w, v_ = v;
        asm volatile("" : "+v"(lane_), "+s"(w_), "+s"(v_));
        const Item it = sk_item(p, v_, idx);
        const SkGroup& g = p.g[it.gi];
        sk_voff<AT, BM>(voffA, w_, lane_, it.m0, p.M, g.lda);
        sk_voff<BT, BN>(voffB, w_, lane_, it.n0, it.Ng, g.ldb);
        const int64_t k0 = (int64_t)it.s0 * SK_BK;
        lA = sk_up(AT ? g.A + it.m0 + k0 * g.lda : g.A + (int64_t)it.m0 * g.lda + k0);
        lB = sk_up(BT ? g.B + it.n0 + k0 * g.ldb : g.B + (int64_t)it.n0 * g.ldb + k0);
        sA = AT ? SK_BK * 4 * g.lda : SK_BK * 4;
        sB = BT ? SK_BK * 4 * g.ldb : SK_BK * 4;
        l_left = it.s1 - it.s0;
    };
    auto l_issue = [&](int pc) {                   // piece pc in [0, PT): A pieces 0 .. PA - 1, then the B pieces of this wave
        if (pc < PA) sk_glds16(lA, (uint32_t)voffA[pc], lds0 + lpar * SK_STAGE + (w * PA + pc) * 1024);
        else sk_glds16(lB, (uint32_t)voffB[pc - PA], lds0 + lpar * SK_STAGE + SK_ABYTES + (w * PB + (pc - PA)) * 1024);
    };
    auto l_advance = [&]() {
        lpar ^= 1;
        if (--l_left > 0) { lA += sA; lB += sB; return; }
        ++li;
        if (li < n_items) l_enter(li);
        else { sA = sB = 0; l_left = 1 << 30; }    // past the end of the stream: stay on the last stage
    };
    l_enter(0);

    // ---- compute cursor
    int ci = 0, c_left = l_left, cpar = 0;

    // prologue: stage 0 whole, first half of stage 1
#pragma unroll
    for (int pc = 0; pc < PT; ++pc) l_issue(pc);
    l_advance();
#pragma unroll
    for (int pc = 0; pc < PT / 2; ++pc) l_issue(pc);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PT / 2) : "memory");       // stage 0 landed, the half of stage 1 stays in flight
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int f = 0; f < TM + TNB; ++f) read_frag(0, sk_lds, 0, f);

    int stage_no = 0;      // (used by the PROBES build only)
    // one stage: q0 .. q3, 16 slots of four MFMAs each, one fragment read or DMA piece pinned behind each.  FAST: the load cursor
    // stays inside its item (plain pointer increments); the other instantiation may enter the next item (rare code in the body)
    auto stage = [&](auto fast_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        const char* st = sk_lds + cpar * SK_STAGE;
        const char* stn = sk_lds + (cpar ^ 1) * SK_STAGE;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cur = q & 1, nxt = cur ^ 1;
            // 4 TM slots of four MFMAs; behind slot s: fragment read s of the next k-group (TM + 4 of them), and in q0 / q3 one
            // of the PT / 2 DMA pieces of that quarter (from the last slot backwards)
#pragma unroll
            for (int s = 0; s < 4 * TM; ++s) {
                const int j = s / TM, i = s % TM;
#pragma unroll
                for (int jn = 0; jn < TNB; ++jn)
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i][j], fb[cur][jn][j], acc[i][jn], 0, 0, 0);
                if (s < TM + TNB) {
                    if (q < 3) read_frag(nxt, st, q + 1, s);
                    else read_frag(nxt, stn, 0, s);              // the next stage's first fragments (unused after the last stage)
                }
                const int d = 4 * TM - 1 - s;                     // DMA piece of this slot
                if (d < PT / 2) {
                    if (q == 0) l_issue(PT / 2 + d);              // second half of the stage after this one
                    else if (q == 3) l_issue(d);                  // first half of the stage after next
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (q == 0) {
                if (FAST) { lpar ^= 1; --l_left; lA += sA; lB += sB; }
                else l_advance();
                __builtin_amdgcn_sched_barrier(0);
            }
            if (q == 2) {
                // this stage's buffer is read out (q3's fragments requested: wait for them), the next stage has landed
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        cpar ^= 1;
        SK_STAGE_STAMP(stage_no); ++stage_no;
    };
    using TagFast = std::true_type;
    using TagSlow = std::false_type;

    for (;;) {
        // the run of stages in which neither cursor changes its item: a loop of nothing but the stage body
        int n = sk_u(min(c_left - 1, l_left - 1));
        c_left -= max(n, 0);
        for (; n > 0; --n) stage(TagFast{});
        stage(TagSlow{});
        if (--c_left > 0) continue;
        SK_STAMP(1 + 2 * ci);
        // ---------------- the item is complete: store it (inputs made opaque: see l_enter)
        int lane_ = lane, w_ = w, v_ = v;
        asm volatile("" : "+v"(lane_), "+s"(w_), "+s"(v_));
        const int lane = lane_, w = w_, v = v_;
        const Item it = sk_item(p, v, ci);
        const SkGroup& g = p.g[it.gi];
        if (it.kind == 1) {
            float* bounce = reinterpret_cast<float*>(sk_lds + 2 * SK_STAGE + w * SK_BOUNCE);
            f32x4 bias4[TNB];
            sk_bias<TNB>(p, g, it, w, lane, bias4);                 // before the first store: a later load would wait behind the stores
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TNB; ++j) {
                    sk_store_block(p, g, it, acc[i][j], i, j, w, lane, bias4[j], bounce);
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
                    __builtin_amdgcn_sched_barrier(0);       // one sub-block's values in flight at a time
                }
        } else {
            // share of a split tile: accumulator order, 16-B stores: slot[((w * NSUB + i * TNB + j) * 4 + e4) * 64 + lane]
            f32x4* slot = reinterpret_cast<f32x4*>(p.slots) + (int64_t)(2 * v + ci) * (SK_SLOT_BYTES / 16);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TNB; ++j) {
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        f32x4 t;
#pragma unroll
                        for (int e = 0; e < 4; ++e) t[e] = acc[i][j][4 * e4 + e];
                        slot[((w * NSUB + i * TNB + j) * 4 + e4) * 64 + lane] = t;
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
                    __builtin_amdgcn_sched_barrier(0);
                }
            // publish: every wave drains its stores, the workgroup meets, one lane releases and counts
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (threadIdx.x == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(p.cnt + 4 * it.r, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        SK_STAMP(2 + 2 * ci);
        if (++ci >= n_items) break;
        const Item nx = sk_item(p, v, ci);
        c_left = nx.s1 - nx.s0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the pieces issued past the end of the stream
    SK_STAMP(27);

    // ---- the split tiles this workgroup contributed to (its first one or two items): contributor c of n adds up the sub-blocks
    // k = c (mod n), shares in contributor order.  R whole tiles lie between a share's store and this point, so the tile is
    // complete here; when it is not (no whole-tile rounds, or the launch is not co-resident: another tenant on the device) the
    // workgroup polls for a bounded few tens of microseconds and then leaves its sub-blocks to the contributor that decides
    // last.  Counters of split tile r: cnt[4r] shares published, [4r + 1] contributors decided, [4r + 2] bit c set =
    // contributor c gave its sub-blocks away.
    const int ns = sk_nsplit(p, v);
    // a decision of thread 0 for the whole workgroup: a word behind the bounce tiles (a static __shared__ variable would be
    // placed in front of the stage buffers and move them off their 1-KB alignment)
    volatile int* sk_flag = reinterpret_cast<volatile int*>(sk_lds + 2 * SK_STAGE + 4 * SK_BOUNCE);      // one word per split item
    float* fbounce = reinterpret_cast<float*>(sk_lds + 2 * SK_STAGE + w * SK_BOUNCE);
    for (int si = 0; si < ns; ++si) {
        const Item S = sk_item(p, v, si);
        if (S.kind != 2) continue;
        int lo, hi;
        sk_contributors(p, S.r, lo, hi);
        const int n = hi - lo + 1, c = v - lo;
        uint32_t* cnt = p.cnt + 4 * S.r;
        const SkGroup& g = p.g[S.gi];
        const bool f_bias = (p.flags & DLSG_GEMM_BIAS) && g.bias != nullptr;
        auto add_up = [&](int c_) {                  // the sub-blocks of contributor c_
            for (int k = c_; k < NSUB; k += n) {
                f32x16 sum;
#pragma unroll
                for (int e = 0; e < 16; ++e) sum[e] = 0.f;
                for (int v2 = lo; v2 <= hi; ++v2) {
                    // contributor v2's share of this tile is its first item when its run begins inside the tile, else its second
                    const int sl = 2 * v2 + ((int)(sk_run0(p, v2) / p.sk_nst) == S.r ? 0 : 1);
                    const f32x4* slot = reinterpret_cast<const f32x4*>(p.slots) + (int64_t)sl * (SK_SLOT_BYTES / 16);
                    f32x4 t[4];
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) t[e4] = slot[((w * NSUB + k) * 4 + e4) * 64 + lane];
#pragma unroll
                    for (int e = 0; e < 16; ++e) sum[e] += t[e >> 2][e & 3];
                }
                const int bcol = S.n0 + (w & 1) * WN + (k % TNB) * 32 + 4 * (lane & 7);
                f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
                if (f_bias && bcol < S.Ng) b4 = *reinterpret_cast<const f32x4*>(g.bias + bcol);
                sk_store_block(p, g, S, sum, k / TNB, k % TNB, w, lane, b4, fbounce);
            }
        };
        if (threadIdx.x == 0) {
            // a short patience (some tens of microseconds: the contributors of a tile whose launch has no whole-tile rounds
            // publish within microseconds of each other), never an open-ended wait
            bool ready = false;
            for (int spin = 0; spin < ((p.flags & DLSG_GEMM_SK_GIVEAWAY) ? 0 : SK_PATIENCE); ++spin) {
                ready = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (uint32_t)n;
                if (ready) break;
                __builtin_amdgcn_s_sleep(8);
            }
            // decided: count this contributor (behind its mask bit when it gives its sub-blocks away).  The last one to decide
            // finds every share published -- a share is published before its owner gets here -- takes what was given away and
            // leaves the counters at zero for the next launch (nobody touches them after its own count).
            uint32_t d;
            if (ready) {
                d = __hip_atomic_fetch_add(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __hip_atomic_fetch_or(cnt + 2, 1u << c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                d = __hip_atomic_fetch_add(cnt + 1, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            uint32_t left = 0;
            if (d + 1 == (uint32_t)n) {
                left = __hip_atomic_load(cnt + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(cnt + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(cnt + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            sk_flag[si] = (ready ? 1 : 0) | (int)(left << 1);
        }
        __syncthreads();
        SK_STAMP(28 + 2 * si);
        const int fl = sk_flag[si];
        if (fl & 1) add_up(c);
        if (fl >> 1) {
            for (int c2 = 0; c2 < n; ++c2)
                if ((fl >> 1) & (1 << c2)) add_up(c2);
        }
        SK_STAMP(29 + 2 * si);
    }
}

int sk_cus_, sk_lds_max_;
void sk_query() {
    static std::once_flag once;
    std::call_once(once, [] {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return;
        (void)hipDeviceGetAttribute(&sk_cus_, hipDeviceAttributeMultiprocessorCount, dev);
        (void)hipDeviceGetAttribute(&sk_lds_max_, hipDeviceAttributeMaxSharedMemoryPerBlock, dev);
    });
}
int sk_cus() { sk_query(); return sk_cus_; }
int sk_lds_max() { sk_query(); return sk_lds_max_; }

constexpr int sk_lds_bytes(int bm, int bn) { return 2 * (bm * 128 + bn * 128) + 4 * SK_BOUNCE + 16; }   // two stage buffers + the epilogue's bounce tiles + a flag

template <int BM, int BN, bool AT, bool BT>
int sk_launch(const SkArgs& k, hipStream_t st) {
    constexpr int lds_bytes = sk_lds_bytes(BM, BN);
    static std::once_flag once;
    static hipError_t attr_rc = hipSuccess;
    std::call_once(once, [] {
        attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_sk_kernel<BM, BN, AT, BT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    });
    if (attr_rc != hipSuccess) return DLSG_ELAUNCH;
    hipLaunchKernelGGL((gemm_sk_kernel<BM, BN, AT, BT>), dim3(k.P), dim3(SK_THREADS), lds_bytes, st, k);
    DLSG_CHECK_LAUNCH();
    return DLSG_OK;
}

}  // namespace

// Workgroups of a launch: one per CU, or the caller's budget (dlsg_gemm_args.cu_budget) rounded down to a multiple of 8 (one share
// per XCD).  A workgroup of this kernel needs a CU to itself (147 KB of LDS, all 512 registers of its SIMDs): when another tenant
// holds CUs -- a collective's kernels on a side stream under the backward -- a launch of one workgroup per CU would run its last
// workgroups only after the first ones left; with a budget that leaves those CUs free it stays one co-resident wave.
static int sk_workgroups(const dlsg_gemm_args* a) {
    int p = sk_cus();
    if (a->cu_budget > 0 && a->cu_budget < p) p = a->cu_budget >= 16 ? (a->cu_budget & ~7) : 8;
    return p;
}

// stages per tile of the first group that owns a tile at or behind linear tile id `from`
static int max_nst_sk(const SkArgs& k, int from, int tiles_m) {
    for (int i = 0; i < k.ngroups; ++i)
        if (k.g[i].tile0 + tiles_m * k.g[i].tiles_n > from) return k.g[i].nst;
    return 0;
}

// bytes of caller scratch a stream-K launch needs on this device (zero-filled once by the caller; the kernels leave the
// counter area zeroed); 0 when the device cannot be queried
extern "C" int64_t dlsg_gemm_ws_bytes(void) {
    const int cus = sk_cus();
    return cus > 0 ? (int64_t)SK_CNT_BYTES + (int64_t)2 * cus * SK_SLOT_BYTES : 0;
}

// 1: this call can run on the stream-K kernel (alignment, K a multiple of 32 in every group, one batch, a workspace)
int dlsg_gemm_sk_ok(const dlsg_gemm_args* a) {
    const bool at = a->mode == 2, bt = a->mode != 0;
    if (a->nbatch != 1 || a->M < 4 || a->N < 4) return 0;
    if (!a->ws || a->ws_bytes < dlsg_gemm_ws_bytes() || dlsg_gemm_ws_bytes() == 0) return 0;
    // what the dispatch relies on, checked HERE so that gemm_plan falls back to the tiled kernels instead of failing the call: the
    // counter area holds four words per split tile (fewer than P of them), a workgroup's stage buffers fit the device's LDS
    if (4 * (sk_workgroups(a) - 1) * (int)sizeof(uint32_t) > 4096) return 0;
    if (sk_lds_max() < sk_lds_bytes(256, 256)) return 0;
    if (at && (a->M % 4)) return 0;
    for (int i = 0; i < a->ngroups; ++i) {
        const dlsg_gemm_group& g = a->g[i];
        const int gn = g.N > 0 ? g.N : a->N;
        if (g.K < SK_BK || (g.K % SK_BK) || (g.lda % 4) || (g.ldb % 4) || gn < 4 || (gn % 4)) return 0;
        if ((reinterpret_cast<uintptr_t>(g.A) & 15) || (reinterpret_cast<uintptr_t>(g.B) & 15)) return 0;
        // 16-B pieces of the result rows (and of the bias)
        const int64_t ldc = g.ldc ? g.ldc : (int64_t)a->ldc;
        const float* bias = g.bias ? g.bias : a->bias;
        if ((reinterpret_cast<uintptr_t>(g.C) & 15) || (ldc % 4)) return 0;
        if ((a->flags & DLSG_GEMM_BIAS) && bias && (reinterpret_cast<uintptr_t>(bias) & 15)) return 0;
        // 32-bit piece offsets inside a tile
        if (!at && (int64_t)255 * g.lda * 4 >= (1LL << 31)) return 0;
        if (at && (int64_t)31 * g.lda * 4 >= (1LL << 31)) return 0;
        if (!bt && (int64_t)255 * g.ldb * 4 >= (1LL << 31)) return 0;
        if (bt && (int64_t)31 * g.ldb * 4 >= (1LL << 31)) return 0;
    }
    return 1;
}

// tiles of the launch on (bm x bn) tiles, and the deepest contraction among its groups
static int64_t sk_count_tiles(const dlsg_gemm_args* a, int bm, int bn, int* kmax) {
    const int64_t tm = (a->M + bm - 1) / bm;
    int64_t tiles = 0;
    int km = 0;
    for (int i = 0; i < a->ngroups; ++i) {
        const int64_t gn = a->g[i].N > 0 ? a->g[i].N : a->N;
        tiles += tm * ((gn + bn - 1) / bn);
        km = a->g[i].K > km ? a->g[i].K : km;
    }
    if (kmax) *kmax = km;
    return tiles;
}
// Tile of a launch (DLSG_GEMM_SK_BM128 / _BM256 / _BN128 force one).  Measured per shape with tools/gemm_sk_probe.py and the
// round's tile table:
//   * 128 rows when 256-row tiles would leave > 3 % of the last row panel empty (M = 1664: 7 panels for 6.5);
//   * 128 rows also when the launch has fewer 256 x 256 tiles than the chip has CUs and its contractions are not deep: every tile
//     is then cut between workgroups, and twice as many tiles mean whole-tile rounds and shares of half the size (TN 2048 x 2048
//     x 1664 x 3: 323 us against 337; the 1024-row weight-gradient group 275 against 292; TN 10 000 x 1536 x 1664: 401 against
//     414).  The deep weight gradients (64 tiles x 832 stages) stay on 256 rows: their fix-up is nothing beside 208 stages;
//   * 128 x 128 only for one-round launches of few, deep tiles (NT 1664 x 1024 x 6144: 171 us against 176): elsewhere the
//     narrower tile is within 3 us of the wider one or of the tiled kernels, either way.
void sk_pick_tile(const dlsg_gemm_args* a, int* bm_, int* bn_) {
    int bm = 256, bn = 256;
    if (a->flags & DLSG_GEMM_SK_BN128) { bm = 128; bn = 128; }
    else if (a->flags & DLSG_GEMM_SK_BM128) bm = 128;
    else if (a->flags & DLSG_GEMM_SK_BM256) bm = 256;
    else {
        const int cus = sk_cus() > 0 ? sk_workgroups(a) : 256;
        const int pad256 = (a->M + 255) / 256 * 256;
        int kmax = 0;
        if ((pad256 - a->M) * 32 > a->M) bm = 128;
        else if (a->M >= 256 && sk_count_tiles(a, 256, 256, &kmax) < cus && kmax <= 8192) bm = 128;
        if (bm == 128 && a->ngroups == 1 && sk_count_tiles(a, 128, 256, &kmax) * 4 <= cus && kmax >= 4096) bn = 128;
    }
    *bm_ = bm; *bn_ = bn;
}

// 1: dlsg_gemm sends this call to the stream-K kernel.  Measured on MI355X against the tiled kernels (tools/gemm_sk_probe.py,
// profiles/r05j_*): it wins wherever a workgroup gets >= ~80 us of stages -- region projections 1 582 us against 1 813, obj_embed
// weight gradients 1 548 against 1 836 (+ their fold), the 1 664-row products of the step by 8-21 % on 128-row tiles (NT
// 1664 x 4096 x 1024 x 2: 230 against 266; NT 1664 x 1024 x 6144: 178 against 227; NN 1664 x 2048 x 2048 x 3: 326 against 354) --
// and loses below that, where the split tiles' fix-up (~30 us) is a third of the launch (NT 1664 x 1024 x 2048: 84 against 80;
// the vocabulary projection 60 against 44).  A launch whose tiles come in whole rounds has no fix-up and pays from ~40 us.
int dlsg_gemm_sk_wanted(const dlsg_gemm_args* a) {
    if (!a->ws || (a->flags & DLSG_GEMM_NOSK) || !dlsg_gemm_sk_ok(a)) return 0;
    if (sk_cus() <= 0) return 0;
    const int cus = sk_workgroups(a);
    int bm, bn;
    sk_pick_tile(a, &bm, &bn);
    const int64_t tm = (a->M + bm - 1) / bm;
    double useful = 0.0, padded = 0.0;
    int64_t units = 0, tiles = 0;
    for (int i = 0; i < a->ngroups; ++i) {
        const int64_t gn = a->g[i].N > 0 ? a->g[i].N : a->N, tn = (gn + bn - 1) / bn;
        tiles += tm * tn;
        units += tm * tn * (a->g[i].K / SK_BK);
        useful += (double)a->M * gn * a->g[i].K;
        padded += (double)tm * bm * tn * bn * a->g[i].K;
    }
    if (useful < 0.85 * padded) return 0;                      // > 15 % of the tiles' area would be padding
    // the tiles that do not fill a whole round are cut between the workgroups only when they are equally deep; otherwise each
    // goes to one workgroup whole (a 6 144-deep and a 2 048-deep group in one launch: 708 us for what takes 181 + 82 apart) --
    // such a launch is left to the tiled kernels unless it is forced here
    if (tiles % cus) {
        int64_t t0 = 0;
        int depth = -1;
        for (int i = 0; i < a->ngroups; ++i) {
            const int64_t gn = a->g[i].N > 0 ? a->g[i].N : a->N, tn = (gn + bn - 1) / bn;
            const int64_t t1 = t0 + tm * tn;
            if (t1 > tiles / cus * cus) {                      // this group owns tiles of the last, partial round
                if (depth >= 0 && depth != a->g[i].K / SK_BK) return 0;
                depth = a->g[i].K / SK_BK;
            }
            t0 = t1;
        }
    }
    const double us = (double)((units + cus - 1) / cus) * 7.0 * bm / 256 * bn / 256;      // a 256 x 256 x 32 stage takes 7.0 us
    // (a launch whose tiles come in whole rounds has no fix-up: TN 1024 x 1024 x 320 x 8 on 128-row tiles is exactly one round,
    //  49.6 us against 61.5 on the small tiles)
    return us >= ((tiles % cus) ? 80.0 : 30.0);
}

int dlsg_gemm_sk_dispatch(const dlsg_gemm_args* a, hipStream_t st) {
    if (!dlsg_gemm_sk_ok(a)) return DLSG_EINVAL;
    SkArgs k;
    k.M = a->M; k.ngroups = a->ngroups; k.flags = a->flags; k.alpha = a->alpha; k.skip_if = a->skip_if;
    // tile height: 128 rows when 256-row tiles would leave the last row panel mostly padding (M = 1664: 7 panels for 6.5) or
    // when the launch has few tiles to deal out; DLSG_GEMM_SK_BM128 / _BM256 force one
    int bm, bn;
    sk_pick_tile(a, &bm, &bn);
    k.bm = bm; k.bn = bn; k.pad3_ = 0;
    const int tiles_m = (a->M + bm - 1) / bm;
    int T = 0;
    for (int i = 0; i < a->ngroups; ++i) {
        const dlsg_gemm_group& g = a->g[i];
        SkGroup& s = k.g[i];
        s.A = g.A; s.B = g.B; s.C = g.C;
        s.bias = g.bias ? g.bias : a->bias;
        s.lda = g.lda; s.ldb = g.ldb; s.ldc = g.ldc ? g.ldc : (int64_t)a->ldc;
        s.N = g.N > 0 ? g.N : a->N;
        s.nst = g.K / SK_BK;
        s.tiles_n = (s.N + bn - 1) / bn;
        s.tile0 = T;
        T += tiles_m * s.tiles_n;
    }
    const int P = sk_workgroups(a);
    k.P = P; k.T = T;
    k.rounds = T / P; k.rem = T % P;
    k.sk_wgs = 0; k.sk_nst = 0; k.xmap = 0;
    if (k.rem > 0) {
        // the last rem tiles: split evenly over the workgroups when their tiles are equally deep (the common case; a run of
        // stages then touches at most two tiles), else one whole tile per workgroup
        bool uniform = true;
        for (int i = 0; i < a->ngroups; ++i)
            if (k.g[i].tile0 + tiles_m * k.g[i].tiles_n > k.rounds * P) uniform = uniform && k.g[i].nst == max_nst_sk(k, k.rounds * P, tiles_m);
        if (uniform) {
            const int nst = max_nst_sk(k, k.rounds * P, tiles_m);
            const int64_t U = (int64_t)k.rem * nst;
            // every workgroup takes part unless that would cut a tile into more than SK_MAXSPLIT - 1 runs (tiny launches)
            int64_t wgs = P;
            if (wgs > (int64_t)k.rem * (SK_MAXSPLIT - 2)) wgs = (int64_t)k.rem * (SK_MAXSPLIT - 2);
            if (wgs > U) wgs = U;
            k.sk_wgs = (int)wgs; k.sk_nst = nst;
            if (k.rounds == 0 && wgs == P && (P & 7) == 0 && P % k.rem == 0 && (P / k.rem == 2 || P / k.rem == 4 || P / k.rem == 8) &&
                nst % (P / k.rem) == 0 && !(a->flags & DLSG_GEMM_SK_NOXMAP))
                k.xmap = P / k.rem;
        } else {
            k.sk_wgs = k.rem; k.sk_nst = 0;
        }
    }
    char* ws = reinterpret_cast<char*>(a->ws);
    k.cnt = reinterpret_cast<uint32_t*>(ws);
    k.slots = reinterpret_cast<float*>(ws + SK_CNT_BYTES);
    k.err = nullptr;
    if (4 * k.rem * (int)sizeof(uint32_t) > 4096) return DLSG_EINVAL;
    switch (a->mode * 3 + (bn == 128 ? 2 : (bm == 128 ? 1 : 0))) {
        case 0: return sk_launch<256, 256, false, false>(k, st);
        case 1: return sk_launch<128, 256, false, false>(k, st);
        case 2: return sk_launch<128, 128, false, false>(k, st);
        case 3: return sk_launch<256, 256, false, true>(k, st);
        case 4: return sk_launch<128, 256, false, true>(k, st);
        case 5: return sk_launch<128, 128, false, true>(k, st);
        case 6: return sk_launch<256, 256, true, true>(k, st);
        case 7: return sk_launch<128, 256, true, true>(k, st);
        case 8: return sk_launch<128, 128, true, true>(k, st);
        default: return DLSG_EINVAL;
    }
}
