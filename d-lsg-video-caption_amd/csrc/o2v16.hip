// Object->frame conditional graph, forward (reference models/layer.py:184-192), second generation (gfx950).
//
//   z[b,t,:] = sum_n softmax_n(scale * o_n . v_t) o_n + v_t,      o_n = obj_norm(y_n)  (LayerNorm, in flight)
//
// The first kernel (attention.hip, 32-object tiles, register staging) serialised HBM round trip -> LayerNorm -> two
// fp32-MFMA phases inside one 148-KB workgroup per CU: ~20 us per tile against a ~7 us matrix-pipe floor.  Here:
//   * tile = 16 objects x H; TWO tile buffers in LDS (2 x 64.3 KB at H = 1024), filled by LDS-DMA
//     (`global_load_lds_dwordx4`: 1-KiB pieces, no staging registers): the loads of tile i+1 are issued before tile i is
//     touched and stay in flight across the barrier inside the tile (raw s_barrier; a __syncthreads() would drain them);
//   * two barriers per tile; the row statistics of tile i+1 (VALU + LDS) are interleaved, slice by slice, with the
//     aggregation MFMAs of tile i, which would otherwise leave the vector pipes idle (phase timings: tools/o2v_stamps.py);
//   * the LayerNorm is never applied to the tile: with o_n = (x_n - mu_n) r_n * gamma + beta,
//         S[n,t]  = scale ( r_n ( x_n . (gamma*v_t) - mu_n (gamma . v_t) ) + beta . v_t )
//         agg_t   = gamma * ( sum_n P r_n x_n - sum_n P r_n mu_n ) + beta sum_n P
//     so both products run on the RAW rows as they landed (gamma folded into the V fragments once per workgroup, the
//     per-frame scalars gamma.v_t, beta.v_t from the prologue) and a wave only computes mean / rstd of the two rows it
//     fetched itself (its own vmcnt orders DMA -> ds_read; DPP + v_readlane reductions): no normalised write-back, no
//     gamma / beta traffic per tile -- with all eight waves in the same phase that pass was LDS-bandwidth bound
//     (5.0k of 19.4k cycles per tile, tools/o2v_stamps.py);
//   * both products on v_mfma_f32_16x16x4_f32 (exact fp32, same rate as 32x32x2):
//       S-product   D[obj][frame] : wave w contracts its H/8 slice of k, V fragments in registers (64 per lane),
//                   8 partial tiles summed through LDS (one barrier: everyone adds all eight);
//       aggregation D[frame][col] : the P registers ARE the A operand (k-step j, lane group g <-> object 4g + j, the
//                   C-layout row order), O from LDS with lanes on consecutive columns;
//   * nsplit == 1 (>= 256 clips in flight): the kernel finishes z = agg / l + v itself, no partial round trip and no
//     combine launch; nsplit > 1 writes flash-decoding partials merged by o2v_combine_kernel (attention.hip).
#include <cstdlib>
#include <mutex>

#include "o2v16.hpp"

using namespace dlsg;
using namespace o16;

namespace {

// STAMP: diagnostic build only (make PROBES=1, i.e. -DDLSG_PROBES, then env DLSG_O2V_STAMPS; tools/o2v_stamps.py): lane 0 of every
// wave of workgroup (0,0) records s_memtime at the phase boundaries of each tile into a.ws (unused when nsplit == 1).  The product
// library neither reads the environment nor contains the stamped instantiation.
// Several graphs of one shape per launch (the object and the motion stream of CapGnnEncoder: 2 x 64 clips fill the chip
// with two object chunks per clip instead of four): blockIdx.z picks the argument block.
struct O16Pack {
    dlsg_o2v_args s[DLSG_O2V_MAXMULTI];
};

template <int H, bool STAMP>
__global__ __launch_bounds__(O16_THREADS) void o2v16_kernel(const O16Pack pk, int tiles_per_split) {
    using G = O16Geom<H>;
    const dlsg_o2v_args& a = pk.s[blockIdx.z];
    auto stamp = [&](int it_, int k_) {
        if constexpr (STAMP) {
            if (blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 63) == 0)
                reinterpret_cast<unsigned long long*>(a.ws)[(it_ * 16 + k_) * 8 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memtime();
        }
    };
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* red = smem + 2 * G::BUF;
    float* gam_l = red + G::RED;                 // obj_norm gamma | beta: epilogue only
    float* bet_l = gam_l + H;
    float* stat_l = bet_l + H;                   // [tile parity][mean[16] | rstd[16]]

    const int b = blockIdx.x, sp = blockIdx.y;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int f = lane & 15, g = lane >> 4;
    const int T = a.T, NO = a.NO;
    const int dbg = STAMP ? (tiles_per_split >> 16) : 0;      // diagnostic build: phases switched off (timing only, wrong results)
    tiles_per_split &= 0xffff;
    const int n_begin = sp * tiles_per_split * O16_TILE;
    const int n_end = min(NO, n_begin + tiles_per_split * O16_TILE);

    // ---- LDS-DMA of one tile: wave w fetches rows 2w, 2w+1 (rows past the end are clamped here and zeroed by the LayerNorm pass)
    // as NP pieces of 64 lanes x VB bytes.  Only the first tile is requested in one go: inside the loop the pieces of tile i+1
    // are issued one by one between the MFMA blocks of tile i -- all 256 workgroups of a launch walk their tiles in step, and
    // 64 KB per CU requested at the same instant is a 17-MB burst that the memory system serves at its peak rate for ~2 us
    // while every wave sits in its DMA issue (4.1 k of 16.8 k cycles per tile, tools/o2v_stamps.py), then idles for 6 us.
    auto issue_piece = [&](int n0, float* dst, int pc) {
        const int rr = pc / G::PPR, q = pc % G::PPR;
        const int row = 2 * w + rr;
        const int n = min(n0 + row, NO - 1);
        const char* rowp = reinterpret_cast<const char*>(a.y + ((int64_t)b * NO + n) * H);
        char* d = reinterpret_cast<char*>(dst + row * G::LDO);
        if constexpr (G::VB == 16) {
            const uint32_t la = __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<uintptr_t>((lds_ptr_t)d)) + q * 1024;
            glds16_asm(uniform_ptr(rowp), (uint32_t)(lane * 16 + q * 1024), la);
        } else {
            glds<G::VB>(rowp + lane * G::VB + q * 64 * G::VB, d + q * 64 * G::VB);
        }
    };
    auto issue_tile = [&](int n0, float* dst) {
#pragma unroll
        for (int pc = 0; pc < G::NP; ++pc) issue_piece(n0, dst, pc);
    };
    if (n_begin < n_end) issue_tile(n_begin, smem);

    for (int j = threadIdx.x; j < H; j += O16_THREADS) { gam_l[j] = a.g_obj[j]; bet_l[j] = a.b_obj[j]; }

    // ---- V fragments (B operand of the S product), gamma folded in: vreg[fb][4c + s] = gamma[k] V[frame 16 fb + f][k],
    //      k = w HS + 16 c + 4 g + s; and this lane's share of gv_t = gamma . v_t, bv_t = beta . v_t
    float vreg[2][G::HS / 4];
    float gv[2] = {0.f, 0.f}, bv[2] = {0.f, 0.f};
    {
        const int k0 = (w % G::KW) * G::HS + 4 * g;
#pragma unroll
        for (int fb = 0; fb < 2; ++fb) {
            const int t = 16 * fb + f;
            const float* vp = a.v + ((int64_t)b * T + min(t, T - 1)) * H + k0;
#pragma unroll
            for (int c = 0; c < G::NCHUNK; ++c) {
                const f32x4 v4 = *reinterpret_cast<const f32x4*>(vp + 16 * c);
                const f32x4 g4 = *reinterpret_cast<const f32x4*>(a.g_obj + k0 + 16 * c);
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.b_obj + k0 + 16 * c);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float vv = (t < T && w < G::KW) ? v4[s] : 0.f;
                    vreg[fb][4 * c + s] = vv * g4[s];
                    gv[fb] += vv * g4[s];
                    bv[fb] += vv * b4[s];
                }
            }
            gv[fb] += __shfl_xor(gv[fb], 16, 64); gv[fb] += __shfl_xor(gv[fb], 32, 64);
            bv[fb] += __shfl_xor(bv[fb], 16, 64); bv[fb] += __shfl_xor(bv[fb], 32, 64);
            if (g == 0) { red[(w * 4 + fb * 2) * 16 + f] = gv[fb]; red[(w * 4 + fb * 2 + 1) * 16 + f] = bv[fb]; }
        }
    }

    f32x4 acc_o[2][G::CBW];
#pragma unroll
    for (int fb = 0; fb < 2; ++fb)
#pragma unroll
        for (int c = 0; c < G::CBW; ++c) acc_o[fb][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f}, c_run[2] = {0.f, 0.f};

    // the V / gamma / beta loads above are done with (the compiler waited for their data), so from here on the only
    // vector-memory loads in flight are LDS-DMA pieces
    lds_barrier();
#pragma unroll
    for (int fb = 0; fb < 2; ++fb) {
        gv[fb] = 0.f; bv[fb] = 0.f;
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) { gv[fb] += red[(ww * 4 + fb * 2) * 16 + f]; bv[fb] += red[(ww * 4 + fb * 2 + 1) * 16 + f]; }
    }
    lds_barrier();      // `red` is reused by the first tile's partial scores

    // ---- row statistics of one tile buffer, for the two rows this wave fetched itself (its own vmcnt ordered DMA -> ds_read),
    // cut into 8 steps so that they can be interleaved with MFMAs: per row {load, sum + reduce, centred squares + reduce,
    // publish}.  DPP + v_readlane reductions, not the LDS crossbar.
    float sx[G::EPL];
    float smean = 0.f, srstd = 0.f;
    auto stats_step = [&](int step, const float* buf, float* st, int n0_) {
        const int rr = step >> 2;
        const int row = 2 * w + rr;
        switch (step & 3) {
            case 0: {
                const float* rp = buf + row * G::LDO;
#pragma unroll
                for (int c = 0; c < G::NCH; ++c) {
                    if (G::VEC == 4) {
                        const f32x4 t4 = *reinterpret_cast<const f32x4*>(rp + c * 256 + 4 * lane);
                        sx[4 * c] = t4[0]; sx[4 * c + 1] = t4[1]; sx[4 * c + 2] = t4[2]; sx[4 * c + 3] = t4[3];
                    } else {
                        sx[c] = rp[c * 64 + lane];
                    }
                }
                break;
            }
            case 1: {
                // (four partial sums: a single 16-deep chain of dependent adds is ~100 cycles in which this wave issues nothing else)
                float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < G::EPL; ++i) s4[i & 3] += sx[i];
                smean = wave_sum_dpp((s4[0] + s4[1]) + (s4[2] + s4[3])) / H;
                break;
            }
            case 2: {
                float q4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < G::EPL; ++i) { const float d = sx[i] - smean; q4[i & 3] += d * d; }
                srstd = rsqrtf(wave_sum_dpp((q4[0] + q4[1]) + (q4[2] + q4[3])) / H + a.eps);
                break;
            }
            default: {
                if (lane == 0) {
                    const int n = n0_ + row;
                    st[row] = smean;
                    st[16 + row] = srstd;
                    if (n < n_end && a.ostats) {
                        a.ostats[2 * ((int64_t)b * NO + n)] = smean;
                        a.ostats[2 * ((int64_t)b * NO + n) + 1] = srstd;
                    }
                }
            }
        }
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the first tile: this wave's rows landed; behind the barrier, everyone's
    lds_barrier();

    int it = 0;
    for (int n0 = n_begin; n0 < n_end; n0 += O16_TILE, ++it) {
        float* cur = smem + (it & 1) * G::BUF;
        float* nxt = smem + ((it + 1) & 1) * G::BUF;
        float* st_cur = stat_l;
        const bool more = n0 + O16_TILE < n_end;
        stamp(it, 0);
        stamp(it, 1);
        stamp(it, 2);
        stamp(it, 3);

        // ---- partial S over this wave's k slice: D[obj 4g+i][frame 16fb+f].  In the MFMA shadows: the first half of the next
        // tile's DMA pieces (`nxt` was released by the barrier that ended the previous tile) and the row statistics of THIS tile
        // for the two rows this wave fetched itself (they landed before that barrier; needed by the softmax behind the next one)
        constexpr int NPH = G::NP / 2;                 // pieces issued in this phase; the rest between the aggregation blocks
        f32x4 sacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        if (w < G::KW) {
            const float* ap = cur + f * G::LDO + w * G::HS + 4 * g;        // A[m = f][k = 16c + 4g + s]
            f32x4 a_nx = *reinterpret_cast<const f32x4*>(ap);
#pragma unroll
            for (int c = 0; c < G::NCHUNK; ++c) {
                // the next chunk's fragment is requested before this chunk's MFMAs (one LDS round trip per chunk otherwise)
                const f32x4 a4 = a_nx;
                if (c + 1 < G::NCHUNK) a_nx = *reinterpret_cast<const f32x4*>(ap + 16 * (c + 1));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) {
                    sacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s_], vreg[0][4 * c + s_], sacc[0], 0, 0, 0);
                    sacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s_], vreg[1][4 * c + s_], sacc[1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (more && !(dbg & 2)) {
#pragma unroll
                    for (int pc = 0; pc < NPH; ++pc)
                        if (pc * G::NCHUNK / NPH == c) issue_piece(n0 + O16_TILE, nxt, pc);
                }
                if (!(dbg & 1)) {
#pragma unroll
                    for (int step = 0; step < 8; ++step)
                        if (step * G::NCHUNK / 8 == c) stats_step(step, cur, st_cur, n0);
                }
            }
            f32x4* r4 = reinterpret_cast<f32x4*>(red);
            r4[(w * 2 + 0) * 64 + lane] = sacc[0];
            r4[(w * 2 + 1) * 64 + lane] = sacc[1];
        } else {
            if (more) {
#pragma unroll
                for (int pc = 0; pc < NPH; ++pc) issue_piece(n0 + O16_TILE, nxt, pc);
            }
#pragma unroll
            for (int step = 0; step < 8; ++step) stats_step(step, cur, st_cur, n0);
        }
        stamp(it, 4);
        lds_barrier();
        stamp(it, 5);
        float p[2][4];
        float alpha[2];
        {
            const f32x4* r4 = reinterpret_cast<const f32x4*>(red);
            const f32x4 mu4 = *reinterpret_cast<const f32x4*>(st_cur + 4 * g);          // objects 4g .. 4g+3
            const f32x4 rs4 = *reinterpret_cast<const f32x4*>(st_cur + 16 + 4 * g);
#pragma unroll
            for (int fb = 0; fb < 2; ++fb) {
                f32x4 sv = r4[fb * 64 + lane];
#pragma unroll
                for (int ww = 1; ww < G::KW; ++ww) sv += r4[(ww * 2 + fb) * 64 + lane];
                const int t = 16 * fb + f;
                float tmax = -INFINITY;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float s_ = a.scale * (rs4[i] * (sv[i] - mu4[i] * gv[fb]) + bv[fb]);
                    const int n = n0 + 4 * g + i;
                    const bool valid = n < n_end;
                    if (w == 0 && valid && t < T && a.S) a.S[((int64_t)b * NO + n) * T + t] = s_;
                    s_ = valid ? s_ : -INFINITY;
                    p[fb][i] = s_;
                    tmax = fmaxf(tmax, s_);
                }
                tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                const float m_new = fmaxf(m_run[fb], tmax);
                alpha[fb] = __expf(m_run[fb] - m_new);          // m_run = -inf on the first tile -> 0
                float psum = 0.f, csum = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float e = __expf(p[fb][i] - m_new);   // invalid rows: exp(-inf) = 0
                    psum += e;
                    p[fb][i] = e * rs4[i];                       // the A operand of the aggregation carries rstd_n
                    csum += p[fb][i] * mu4[i];
                }
                // this lane's 4 objects only: the 4 lane groups are added once, after the loop
                l_run[fb] = l_run[fb] * alpha[fb] + psum;
                c_run[fb] = c_run[fb] * alpha[fb] + csum;
                m_run[fb] = m_new;
            }
        }
        stamp(it, 6);
        // ---- aggregation: acc_o[fb][c] (frame 16fb+4g+i, col 16cb+f) = alpha_frame * acc_o + sum_n P[n][frame] O[n][col],
        //      with the statistics of the NEXT tile (whose DMA was issued a whole S-product ago) in the MFMA shadows
#pragma unroll
        for (int fb = 0; fb < 2; ++fb) {
            float arow[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) arow[i] = __shfl(alpha[fb], 4 * g + i, 64);
#pragma unroll
            for (int c = 0; c < G::CBW; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc_o[fb][c][i] *= arow[i];
        }
        float b_nx[4];
        {
            const float* bp = cur + min(w * G::CBW, G::NCB - 1) * 16 + f;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) b_nx[jj] = bp[(4 * g + jj) * G::LDO];
        }
#pragma unroll
        for (int c = 0; c < G::CBW; ++c) {
            const int cb = w * G::CBW + c;
            float bvv[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) bvv[jj] = b_nx[jj];
            if (c + 1 < G::CBW) {
                const float* bp = cur + min(cb + 1, G::NCB - 1) * 16 + f;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) b_nx[jj] = bp[(4 * g + jj) * G::LDO];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (cb < G::NCB) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    acc_o[0][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[0][jj], bvv[jj], acc_o[0][c], 0, 0, 0);
                    acc_o[1][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[1][jj], bvv[jj], acc_o[1][c], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (more && !(dbg & 2)) {
#pragma unroll
                for (int pc = NPH; pc < G::NP; ++pc)
                    if ((pc - NPH) * G::CBW / (G::NP - NPH) == c) issue_piece(n0 + O16_TILE, nxt, pc);
            }
        }
        if (more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's rows of the next tile landed
        stamp(it, 7);
        lds_barrier();      // `cur` is free for the DMA of tile i+2, `red` for the next partials, tile i+1's statistics are out
        stamp(it, 8);
    }

#pragma unroll
    for (int fb = 0; fb < 2; ++fb) {
        l_run[fb] += __shfl_xor(l_run[fb], 16, 64);
        l_run[fb] += __shfl_xor(l_run[fb], 32, 64);
        c_run[fb] += __shfl_xor(c_run[fb], 16, 64);
        c_run[fb] += __shfl_xor(c_run[fb], 32, 64);
    }
    // agg[t][col] = gamma[col] (acc[t][col] - c_t) + beta[col] l_t   (un-normalised: divided by the merged l at the end)
    float gcol[G::CBW], bcol[G::CBW];
#pragma unroll
    for (int c = 0; c < G::CBW; ++c) {
        const int cb = min(w * G::CBW + c, G::NCB - 1);
        gcol[c] = gam_l[cb * 16 + f];
        bcol[c] = bet_l[cb * 16 + f];
    }
    if (a.nsplit == 1) {
        // ---- finish in place: z = agg / l + v.  All residual loads first (clamped rows, no branches), then the stores: a
        // load -> wait -> store chain per element would be 64 serial memory round trips.
        float res[2][G::CBW][4];
#pragma unroll
        for (int fb = 0; fb < 2; ++fb)
#pragma unroll
            for (int c = 0; c < G::CBW; ++c) {
                const int cb = min(w * G::CBW + c, G::NCB - 1);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int t = min(16 * fb + 4 * g + i, T - 1);
                    res[fb][c][i] = a.v[((int64_t)b * T + t) * H + cb * 16 + f];
                }
            }
#pragma unroll
        for (int fb = 0; fb < 2; ++fb) {
            float linv[4], crow[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                linv[i] = 1.f / __shfl(l_run[fb], 4 * g + i, 64);
                crow[i] = __shfl(c_run[fb], 4 * g + i, 64);
            }
#pragma unroll
            for (int c = 0; c < G::CBW; ++c) {
                const int cb = w * G::CBW + c;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int t = 16 * fb + 4 * g + i;
                    if (t < T && cb < G::NCB)
                        a.z[((int64_t)b * T + t) * H + cb * 16 + f] =
                            gcol[c] * (acc_o[fb][c][i] - crow[i]) * linv[i] + bcol[c] + res[fb][c][i];
                }
            }
            const int t = 16 * fb + f;
            if (w == 0 && g == 0 && t < T && a.ml) {
                a.ml[2 * ((int64_t)b * T + t)] = m_run[fb];
                a.ml[2 * ((int64_t)b * T + t) + 1] = l_run[fb];
            }
        }
        return;
    }
    // ---- partial results: ws[(b*nsplit+sp)] = { agg[T][H], m[32], l[32] }
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
            wsp[(int64_t)T * H + t] = m_run[fb];
            wsp[(int64_t)T * H + 32 + t] = l_run[fb];
        }
    }
}

template <int H>
int o2v16_launch_t(const dlsg_o2v_args* a, int count, hipStream_t st) {
    using G = O16Geom<H>;
    static std::once_flag once;
    constexpr int lds_bytes = G::LDS_FLOATS * 4;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&o2v16_kernel<H, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  lds_bytes);
#ifdef DLSG_PROBES
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&o2v16_kernel<H, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  lds_bytes);
#endif
    });
    const int tiles = (a->NO + O16_TILE - 1) / O16_TILE;
    const int tps = (tiles + a->nsplit - 1) / a->nsplit;
    O16Pack pk;
    for (int i = 0; i < count; ++i) pk.s[i] = a[i];
    const dim3 grid(a->B, a->nsplit, count);
#ifdef DLSG_PROBES
    static const bool stamps = getenv("DLSG_O2V_STAMPS") != nullptr;
    if (stamps && a->nsplit == 1) {
        const char* d = getenv("DLSG_O2V_DBG");
        hipLaunchKernelGGL((o2v16_kernel<H, true>), grid, dim3(O16_THREADS), lds_bytes, st, pk, tps | ((d ? atoi(d) : 0) << 16));
        return hipGetLastError() == hipSuccess ? DLSG_OK : DLSG_ELAUNCH;
    }
#endif
    hipLaunchKernelGGL((o2v16_kernel<H, false>), grid, dim3(O16_THREADS), lds_bytes, st, pk, tps);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? DLSG_OK : DLSG_ELAUNCH;
}

}  // namespace

// called by dlsg_o2v_fwd / dlsg_o2v_fwd_multi (attention.hip); the combine launch for nsplit > 1 stays there
int dlsg_o2v16_partial(const dlsg_o2v_args* a, int count, hipStream_t st) {
    switch (a->H) {
        case 1024: return o2v16_launch_t<1024>(a, count, st);
        case 512: return o2v16_launch_t<512>(a, count, st);
        case 64: return o2v16_launch_t<64>(a, count, st);
        default: return DLSG_EINVAL;
    }
}
