/* dlsg.h -- C ABI of libdlsg_hip.so: the MI355X (gfx950) kernels under the D-LSG CapGnnModel hot path.
 *
 * The reference (baiyang4/D-LSG-Video-Caption) has no native layer: every device op is a stock torch.nn call.
 * Each entry point below replaces the stock op(s) at the cited reference location.  All pointers are device
 * pointers owned by the caller; the library allocates nothing, keeps no global state, launches asynchronously
 * on `stream` (a hipStream_t passed as void*) and returns 0 or a negative DLSG_E* code.  All tensors are dense
 * fp32 row-major unless a leading dimension is given; ids are int64.
 *
 * Host-side bindings: d-lsg-video-caption_amd/dlsg_amd/hip.py (ctypes).  See INTEGRATION.md.
 */
#ifndef DLSG_H
#define DLSG_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever the signature or the meaning of an existing entry point changes (2: dlsg_colsum / dlsg_colsum2 take a
 * workspace pointer before the stream; the RCCL communicator entry points and the persistent BiLSTM were added).  A binding
 * must refuse a library whose version differs from the header it was written against.  (4: the critic's per-op entry points --
 * dlsg_lstm_cell_*, dlsg_tanh_ln_*, dlsg_conv_taps, dlsg_softmax_bwd2, dlsg_gemm_narrow -- gave way to the blocks of its schedule,
 * dlsg_crit_* / dlsg_cln_*; dlsg_lstm_seq takes batch-major arrays.  5: dlsg_gemm_args carries a workspace and an error word for the
 * stream-K kernel, dlsg_gemm_ws_bytes added; dlsg_adam takes `guard` before the stream and dlsg_select_embed `prefilled`.
 * 6: dlsg_gemm_args.err reserved, the stream-K kernel no longer waits.  7: dlsg_dec_tail_args samples the next word in the launch
 * (`s_*`), dlsg_o2v_bwd_args.dysum.  8: dlsg_gemm_args.cu_budget (was padding), dlsg_comm_rehearsal.) */
#define DLSG_ABI_VERSION 8
int dlsg_abi_version(void);

/* Return codes of every entry point that returns int: 0 or one of these. */
#define DLSG_OK 0
#define DLSG_EINVAL (-1)  /* an argument is out of the supported range (shape, count, NULL where data is required) */
#define DLSG_ELAUNCH (-2) /* the HIP runtime refused the launch / an RCCL call failed */
#define DLSG_EALIGN (-3)  /* a pointer or leading dimension misses the alignment the vector loads need */
#define DLSG_ENOCOMM (-4) /* librccl could not be loaded or the communicator handle is invalid */
/* sizeof() of the i-th argument struct below (0 gemm_args, 1 rowln_args, 2 rowln_bwd_args, 3 o2v_args, 4 decatt_args,
 * 5 decatt_bwd_args, 6 lstm_pw_args, 7 lstm_pw_bwd_args, 8 dec_mid_args, 9 dec_tail_args, 10 dec_mid_bwd_args, 11 decatt_cache_grads_args,
 * 12 o2v_bwd_args, 13 latent_psl_args, 14 sa_core_args, 15 beam_select_args,
 * 16 gather_multi_args, 17 sa_core_bwd_args, 18 latent_psl_bwd_args, 19 bilstm_args, 20 bilstm_bwd_args, 21 colsum_desc,
 * 22 lstm_seq_args, 23 cln_args, 24 crit_sa_args, 25 crit_pattn_args, 26 crit_tsum_args, 27 crit_score_args, 28 crit_colsum_desc,
 * 29 crit_reduce_desc): lets a binding verify its struct layout without a GPU. */
int dlsg_struct_size(int which);

/* ---------------------------------------------------------------- GEMM (fp32-in / fp32-acc MFMA 32x32x2)
 * Replaces every nn.Linear / LSTM gate matmul / torch.matmul on the path and their backward products:
 *   obj_embed layer.py:184, visual_embed :179, linear_embed :51, nn.LSTM :52, SelfAttention K/Q/V/out
 *   sublayer.py:66-68,80, AttentionShare K/V/Q/out sublayer.py:29-31,41, LSTMCell layer.py:571,593,
 *   word_restore layer.py:600.
 * C_g[b] (M x N, ldc) = alpha * opA(A_g[b]) . opB(B_g[b])  (+ bias[n]) (+ C) (tanh)   for every group g, batch b
 *   mode 0 "NT": A[m*lda+k], B[n*ldb+k]      (y = x W^T, Linear forward)
 *   mode 1 "NN": A[m*lda+k], B[k*ldb+n]      (dx = dy W)
 *   mode 2 "TN": A[k*lda+m], B[k*ldb+n]      (dW = dy^T x)
 * Groups share M, N, mode and flags but have their own operands, K and output: a K-split or a sum of several
 * (activation, weight) segments is expressed as groups writing separate slabs that the consumer kernel sums.
 */
#define DLSG_GEMM_MAXG 16
#define DLSG_GEMM_ACCUM 1 /* C += result */
#define DLSG_GEMM_BIAS 2  /* + bias[n]   */
#define DLSG_GEMM_TANH 4  /* tanh(.)     */
#define DLSG_GEMM_FORCE64 256  /* tuning: force the 64x64 block tile  */
#define DLSG_GEMM_FORCE128 512 /* tuning: force the 128x128 block tile (both FORCE bits: the 128x64 tile) */
#define DLSG_GEMM_TILE256 2048  /* tuning: force the 256x256 block tile of csrc/gemm_big.hip (with FORCE128: 256x128); EINVAL when
                                   the operands do not meet its alignment conditions */
#define DLSG_GEMM_SK 4096      /* tuning: force the persistent stream-K kernel of csrc/gemm_sk.hip; EINVAL without a workspace
                                  or when the operands do not meet its conditions (16-B alignment, K % 32 == 0, one batch) */
#define DLSG_GEMM_NOSK 8192    /* tuning: never the stream-K kernel */
#define DLSG_GEMM_SK_BM128 16384 /* tuning: the stream-K kernel on 128 x 256 tiles */
#define DLSG_GEMM_SK_BM256 32768 /* tuning: the stream-K kernel on 256 x 256 tiles */
#define DLSG_GEMM_SK_BN128 262144 /* tuning: the stream-K kernel on 128 x 128 tiles (shares half the size of the 128 x 256 tile's) */
#define DLSG_GEMM_SK_NOXMAP 131072 /* tuning: the stream-K kernel keeps all contributors of a tile on one XCD (no K-slice per XCD map) */
#define DLSG_GEMM_SK_GIVEAWAY 65536 /* test hook: no contributor of a split tile finds it complete -- every one gives its
                                       sub-blocks to the contributor that decides last (the path a launch that is not
                                       co-resident takes); same result bit for bit */
#define DLSG_GEMM_BF16X3 1024  /* split-bf16 matrix path: x = hi + lo, 3 bf16 MFMAs per product, fp32 accumulate
                                  (~1e-5 relative error per product instead of 6e-8; see csrc/gemm_bf16x3.hip) */
typedef struct {
    const float* A;
    const float* B;
    float* C;
    int64_t lda, ldb;
    int32_t K;
    int32_t N;   /* 0 = use args.N; otherwise this group's own output width (<= args.N): lets one launch write
                    column blocks of different widths (e.g. dG.W_ih and dG.W_hh) */
    const float* bias; /* optional per-group bias (overrides args.bias when DLSG_GEMM_BIAS is set) */
    int64_t ldc;       /* 0 = args.ldc; otherwise this group's own output row stride (groups writing into different arrays,
                          e.g. the weight-gradient blocks of one LSTM cell in one launch) */
} dlsg_gemm_group;
typedef struct {
    int32_t mode, M, N, ldc;
    int32_t ngroups, nbatch, flags;
    int32_t cu_budget;     /* 0 = the stream-K kernel launches one workgroup per CU; n > 0: at most n (rounded down to a multiple of
                              8), leaving the other CUs to a co-tenant -- a collective's kernels on another stream.  A stream-K
                              workgroup needs a whole CU, so without the budget its last workgroups would start only when the
                              first ones leave.  The tiled kernels ignore it */
    int64_t bsa, bsb, bsc; /* batch strides in elements */
    float alpha;
    int32_t pad2_;
    const float* bias;
    const int32_t* skip_if; /* optional device flag: the launch does nothing when *skip_if != 0 (a hipGraph-replayed step whose
                               product is needed only on some replays, e.g. per-word logits under scheduled sampling) */
    void* ws;               /* optional caller scratch of >= dlsg_gemm_ws_bytes() bytes, 16-B aligned, zero-filled ONCE by the
                               caller and not shared by launches that may run concurrently (one per stream): with it the
                               chip-filling products run on the persistent stream-K kernel (csrc/gemm_sk.hip), which leaves
                               the counter area at the front of it zeroed again; NULL = the tiled kernels only */
    int64_t ws_bytes;
    int32_t* err;           /* reserved (NULL): the stream-K kernel has no inter-workgroup wait that could time out -- the
                               contributor that completes a split tile adds the shares up */
    dlsg_gemm_group g[DLSG_GEMM_MAXG];
} dlsg_gemm_args;
int dlsg_gemm(const dlsg_gemm_args* args, void* stream);
/* Scratch a stream-K launch needs on the current device: a 16-KB counter area + two 256-KB accumulator slots per CU. */
int64_t dlsg_gemm_ws_bytes(void);
/* The kernel family dlsg_gemm runs this call on (fp32 arithmetic; EINVAL with DLSG_GEMM_BF16X3): what a profiler's kernel
 * symbol will be, without repeating the dispatch rule on the caller's side. */
#define DLSG_GEMM_V_64 0        /* gemm_kernel<64, 64, ...> */
#define DLSG_GEMM_V_128x64 1    /* gemm_kernel_w3<128, 64, ...> */
#define DLSG_GEMM_V_128 2       /* gemm_kernel_w3<128, 128, ...> */
#define DLSG_GEMM_V_SKINNY 3    /* skinny_kernel / skinny2_nt_kernel (M <= 128) */
#define DLSG_GEMM_V_256 4       /* gemm_big_kernel<256, 256, ...> */
#define DLSG_GEMM_V_256x128 5   /* gemm_big_kernel<256, 128, ...> */
#define DLSG_GEMM_V_256_HEAD 6  /* gemm_big_kernel<256, 256, ...> on the row panels that come in whole rounds of the CUs, the
                                   remaining rows through the choice again */
#define DLSG_GEMM_V_SK 7        /* gemm_sk_kernel<...>: one persistent launch, 256 x 256 tiles, stream-K remainder */
int dlsg_gemm_variant(const dlsg_gemm_args* args);

/* out[r, :] = sum_s slabs[s][r, :] (+ bias) (tanh); slabs are nslab consecutive (rows x n) arrays. */
int dlsg_slab_reduce(const float* slabs, int nslab, int64_t slab_stride, const float* bias, float* out,
                     int64_t rows, int n, int ldo, int flags, void* stream);

/* ---------------------------------------------------------------- row kernels: (tanh) -> LayerNorm -> (tanh) (+pe) (dropout)
 * Replaces nn.Sequential(Tanh, LayerNorm[, Dropout]) at layer.py:145-163, sublayer.py:183-187,21-26, and
 * the bare LayerNorms at layer.py:53,57,574,599 (+ PositionalEncoding_old add sublayer.py:102-104).
 *   z = x (+ res);  t = pre_tanh ? tanh(z) : z;  y = LN(t)*gamma+beta;  y = post_tanh ? tanh(y) : y;
 *   y = drop1(y);  if pe: y = drop2(y + pe[row % pe_rows]).   stats[row] = {mean, rstd}.
 */
typedef struct {
    const float* x; int64_t ldx;
    const float* res; int64_t ldres;      /* optional residual (NULL = none) */
    const float* gamma; const float* beta;
    float* y; int64_t ldy;
    float* stats;                          /* rows x 2, optional */
    const float* pe; int32_t pe_rows;     /* optional positional table (pe_rows x n) */
    int32_t rows, n;
    int32_t pre_tanh, post_tanh;
    float eps;
    float p1, p2;                          /* dropout probs (0 = off) */
    uint64_t seed; uint32_t site1, site2;
    const uint64_t* seed_ptr;              /* optional device word added to `seed` (graph replay: new masks per replay) */
} dlsg_rowln_args;
int dlsg_rowln_fwd(const dlsg_rowln_args* a, void* stream);
/* `count` (<= DLSG_ROWLN_MAXMULTI) norms of the same number of rows in one launch (the two encoder streams' visual_norm /
 * obj_visual_norm: 1 664 rows each put half the chip to work) */
#define DLSG_ROWLN_MAXMULTI 2
int dlsg_rowln_fwd_multi(const dlsg_rowln_args* a, int count, void* stream);
/* backward: dy -> dx (same shape as x; residual gets the same gradient), dgamma/dbeta partial sums are written
 * to dgb_part (nblk x 2 x n) and folded by dlsg_colsum. */
typedef struct {
    dlsg_rowln_args f;                     /* same descriptor as forward (y unused) */
    const float* dy; int64_t lddy;
    float* dx; int64_t lddx;
    int32_t accum_dx;                      /* dx += instead of = */
    float* dgb_part; int32_t nblk;         /* workspace: nblk x 2 x n ; nblk = grid size used */
} dlsg_rowln_bwd_args;
int dlsg_rowln_bwd(const dlsg_rowln_bwd_args* a, void* stream);
int dlsg_rowln_bwd_multi(const dlsg_rowln_bwd_args* a, int count, void* stream);
int dlsg_rowln_bwd_nblk(int rows);
/* out[j] (+)= sum_r part[r*ld + j], r < rows.  Tall inputs (rows >= 4096, e.g. the 26 624-row bias gradient of the region
 * projection) are summed in row chunks: with ws (>= dlsg_colsum_ws_floats(rows, n) floats of caller scratch) the chunk
 * partials are combined in a fixed order (bit-reproducible), with ws == NULL by float atomics. */
int64_t dlsg_colsum_ws_floats(int rows, int n);
int dlsg_colsum(const float* part, int64_t ld, int rows, int n, float* out, int accum, float* ws, void* stream);
/* two destinations in one pass: dup == 0 -> columns [0,split) to out_a, [split,n) to out_b (gamma | beta halves of the
 * LayerNorm partials); dup != 0 -> all n columns to both (bias_ih / bias_hh of an LSTM receive the same gradient). */
int dlsg_colsum2(const float* part, int64_t ld, int rows, int n, float* out_a, float* out_b, int split, int dup, int accum,
                 float* ws, void* stream);
/* `count` (<= DLSG_COLSUM_MAXMULTI) short column sums in one launch: each descriptor is one dlsg_colsum (out_b NULL) or
 * dlsg_colsum2 (out_b set; dup != 0: every column to both destinations, else columns [0, split) to out_a and the rest to out_b).
 * Only inputs dlsg_colsum_multi_ok() accepts (fewer than 4096 rows, 16-byte aligned, n and ld multiples of 4); no two descriptors of
 * one call may write the same destination.  Same additions in the same order as the single launches. */
#define DLSG_COLSUM_MAXMULTI 32
typedef struct {
    const float* part; int64_t ld;
    float* out_a; float* out_b;
    int32_t rows, n, split, dup, accum, pad_;
} dlsg_colsum_desc;
int dlsg_colsum_multi_ok(const float* part, int64_t ld, int rows, int n);
int dlsg_colsum_multi(const dlsg_colsum_desc* d, int count, void* stream);

/* ---------------------------------------------------------------- object->frame conditional graph (layer.py:184-193)
 * y (B, NO, H) = tanh(obj_embed(regions)) (the GEMM epilogue applied tanh); LayerNorm(obj_norm) is applied on the
 * fly.  v (B, T, H) = visual_norm output.  Computes
 *   S[n,t] = LN(y_n).v_t / sqrt(obj_size);  P = softmax over n;  agg_t = sum_n P[n,t] LN(y_n);  z = agg + v
 * Partial kernel: grid (B, nsplit) online-softmax over an object chunk; combine merges chunks, writes z (pre
 * obj_visual_norm), the softmax statistics m,l (B,T) and the object LN stats (B*NO x 2).
 */
typedef struct {
    const float* y; const float* v;
    const float* g_obj; const float* b_obj;     /* obj_norm.1 weight/bias */
    float* z;                                    /* (B,T,H) agg + v */
    float* ml;                                   /* (B,T,2) softmax max / sum (of exp(S-max)) */
    float* ostats;                               /* (B*NO,2) LN mean/rstd of y rows */
    float* S;                                    /* (B,NO,T) raw scaled scores, saved for backward */
    float* ws; int64_t ws_bytes;                 /* workspace for partials */
    int32_t B, T, NO, H, nsplit;
    float scale, eps;
} dlsg_o2v_args;
int64_t dlsg_o2v_workspace_bytes(int B, int T, int H, int nsplit);
int dlsg_o2v_fwd(const dlsg_o2v_args* a, void* stream);
/* `count` (<= DLSG_O2V_MAXMULTI) graphs of one shape (B, T, NO, H, nsplit equal) in one launch: CapGnnEncoder runs the same
 * graph on the object and the motion stream (models/model.py:69-73); together they fill the chip with half the object
 * chunks per clip.  a[0..count) are consecutive argument blocks. */
#define DLSG_O2V_MAXMULTI 2
int dlsg_o2v_fwd_multi(const dlsg_o2v_args* a, int count, void* stream);
/* Backward of the fused graph in two passes over y (csrc/o2v16_bwd.hip; one workgroup per (clip, object chunk, stream) on
 * the forward's LDS-DMA tile pipeline): given dz (B,T,H) it writes
 *   dy (B,NO,H)  grad wrt the obj_embed pre-activation (through obj_norm's LayerNorm and the tanh of the GEMM epilogue),
 *   dv (B,T,H)   grad wrt the frame nodes v (includes the residual dz),
 *   part (B*nsplit,2,H) dgamma | dbeta of obj_norm per (clip, object chunk) (fold with dlsg_colsum2),
 *   dysum (B*nsplit,H), optional: the column sums of dy per (clip, object chunk) -- obj_embed's bias gradient without another
 *                pass over the (B*NO)-row dy (fold with dlsg_colsum).
 * y, ostats, S, ml, z are the forward's inputs / outputs; pd (B,NO,64) and m12 (B,NO,2) are workspaces, ws = dlsg_o2v_workspace_bytes
 * (B, T, H, nsplit) bytes of chunk partials of dv (needed when nsplit > 1).  dlsg_o2v_bwd_multi: `count` (<= DLSG_O2V_MAXMULTI)
 * graphs of one shape in one launch per pass (the object and the motion stream of CapGnnEncoder).
 * Same support as the forward (T <= 32, H in {64,512,1024}); otherwise the caller runs the unfused chain
 * (dlsg_softmax_fwd/bwd on S + batched dlsg_gemm + dlsg_rowln_bwd; engine.py tun_bwd).
 */
typedef struct {
    const float* y; const float* ostats; const float* g_obj; const float* b_obj;
    const float* v; const float* z; const float* dz; const float* S; const float* ml;
    float* pd; float* m12;
    float* dy; float* dv; float* part;
    float* ws; int64_t ws_bytes;
    int32_t B, T, NO, H, nsplit;
    float scale;
    float* dysum;
} dlsg_o2v_bwd_args;
int dlsg_o2v_bwd_multi(const dlsg_o2v_bwd_args* a, int count, void* stream);
int dlsg_o2v_bwd(const dlsg_o2v_bwd_args* a, void* stream);

/* ---------------------------------------------------------------- LatentPSL forward (sublayer.py:189-198), one launch
 * adj (B,T,P) = softmax over the frames of ov . theta^T;  u (B*P,H) = adj^T ov;  out = Dropout(LayerNorm(tanh(u))).
 * T <= 32, P <= 32, H <= 2048 (multiple of 4), T*H*4 <= 140 KB of LDS; otherwise the caller runs the unfused chain
 * (2 x dlsg_gemm + dlsg_softmax_fwd + dlsg_rowln_fwd), which is also what the backward consumes (adj, u, stats). */
typedef struct {
    const float* ov; const float* theta; const float* gamma; const float* beta;
    float* adj; float* u; float* out; float* stats;
    int32_t B, T, P, H;
    float p; uint32_t site; float eps; float pad_;
    uint64_t seed; const uint64_t* seed_ptr;
} dlsg_latent_psl_args;
int dlsg_latent_psl_fwd(const dlsg_latent_psl_args* a, void* stream);
/* `count` (<= DLSG_PSL_MAXMULTI) LatentPSL modules of one shape (B, P, H equal) in one launch: CapGnnEncoder's object and
 * motion stream (one workgroup per clip: 64 clips alone leave three quarters of the chip idle) */
#define DLSG_PSL_MAXMULTI 2
int dlsg_latent_psl_fwd_multi(const dlsg_latent_psl_args* a, int count, void* stream);
/* backward (P <= 8, (T+8)*H*4 <= 150 KB of LDS): dout (B*P,H) -> dov (B*T,H) written, dtheta_part (B,P,H) and
 * part (B,2,H) = per-clip partials of dtheta and of out_norm's dgamma | dbeta (fold with dlsg_colsum / dlsg_colsum2).
 * u, stats, adj are the forward's outputs; p/site/seed the forward's dropout. */
typedef struct {
    const float* dout; const float* u; const float* stats; const float* gamma;
    const float* adj; const float* ov; const float* theta;
    float* dov; float* dtheta_part; float* part;
    int32_t B, T, P, H;
    float p; uint32_t site;
    uint64_t seed; const uint64_t* seed_ptr;
} dlsg_latent_psl_bwd_args;
int dlsg_latent_psl_bwd(const dlsg_latent_psl_bwd_args* a, void* stream);
int dlsg_latent_psl_bwd_multi(const dlsg_latent_psl_bwd_args* a, int count, void* stream);

/* ---------------------------------------------------------------- SelfAttention 26x26 core (sublayer.py:69-78), one launch
 * w (B,T,T) = softmax_j(K_i . Q_j * scale) (optional mask (B,T,T): mask <= 0 -> -9e15 as sublayer.py:70-72);
 * out (B,T,D) = w V.  T <= 32, D a multiple of 64; otherwise dlsg_gemm + dlsg_softmax_fwd + dlsg_gemm. */
typedef struct {
    const float* K; const float* Q; const float* V; const float* mask;
    float* w; float* out;
    int32_t B, T, D; float scale;
} dlsg_sa_core_args;
int dlsg_sa_core_fwd(const dlsg_sa_core_args* a, void* stream);
/* backward of the core: dout (B,T,D) and the saved w -> dK, dQ, dV (B,T,D), all dense (row stride D). */
typedef struct {
    const float* w; const float* K; const float* Q; const float* V; const float* dout;
    float* dK; float* dQ; float* dV;
    int32_t B, T, D; float scale;
} dlsg_sa_core_bwd_args;
int dlsg_sa_core_bwd(const dlsg_sa_core_bwd_args* a, void* stream);

/* ---------------------------------------------------------------- softmax along the middle axis of (outer, n, inner)
 * LatentPSL softmax over frames (sublayer.py:192: outer=B, n=T, inner=P), SelfAttention row softmax
 * (sublayer.py:74: inner=1, optional mask: mask<=0 -> -9e15 as sublayer.py:70-72), o2v softmax over objects
 * (layer.py:188: n=T*O, inner=T).  The dense products around them run through dlsg_gemm (batched). */
int dlsg_softmax_fwd(const float* x, const float* mask, float* y, int64_t outer, int n, int inner, void* stream);
/* dx = y * (dy - sum_n y*dy) */
int dlsg_softmax_bwd(const float* y, const float* dy, float* dx, int64_t outer, int n, int inner, void* stream);

/* ---------------------------------------------------------------- decoder attention over cached K', V' (sublayer.py:28-43)
 * For stream s in {0,1}: score_p = K'_s[b,p,:].q[b,:] * scale; w = softmax over p; c = sum_p w_p V'_s[b,p,:].
 * K' = (m W_K^T) W_Q and V' = (m W_V^T) W_O^T are precomputed once per forward (step-invariant), so the per-step
 * work is this kernel only.  Writes c (pre tanh/LN) and alpha (B, 2P). */
typedef struct {
    const float* Kp[2]; const float* Vp[2];   /* (B,P,Q) and (B,P,H) */
    const float* q; int64_t ldq;              /* (B,Q) */
    float* c[2]; int64_t ldc;                 /* (B,H) each */
    float* alpha;                             /* (B, nstream*P) */
    int32_t B, P, Q, H, nstream;
    float scale;
} dlsg_decatt_args;
int dlsg_decatt_fwd(const dlsg_decatt_args* a, void* stream);
typedef struct {
    dlsg_decatt_args f;
    const float* dc[2]; int64_t lddc;         /* (B,H) */
    const float* dalpha;                      /* optional (B, nstream*P) */
    float* dKp[2]; float* dVp[2];             /* accumulated (+=) over steps */
    float* dq; int64_t lddq; int32_t accum_dq;
} dlsg_decatt_bwd_args;
int dlsg_decatt_bwd(const dlsg_decatt_bwd_args* a, void* stream);

/* ---------------------------------------------------------------- fused decoder step (Decoder.decode, layer.py:569-602)
 * One workgroup per batch row; everything between the two gate GEMMs of a word step in ONE launch:
 *   query LSTM cell pointwise (sums the gate GEMM's slabs) -> query_lstm_layernorm (+dropout) -> attention over the
 *   cached K', V' of every stream (scores, softmax over P, weighted V') -> tanh -> output_layer LayerNorm (+dropout).
 * Replaces dlsg_lstm_pw_fwd + dlsg_rowln_fwd + dlsg_decatt_fwd + 2 x dlsg_rowln_fwd of the unfused schedule. */
typedef struct {
    /* query cell */
    const float* slabs; int32_t nslab; int32_t pad_; int64_t slab_stride;     /* (S,B,4Q) gate partial sums */
    const float* addend; int64_t ldadd;          /* (B,4Q) global-feature gate part */
    const float* b_ih; const float* b_hh;
    const float* c_prev;                         /* (B,Q) */
    float* c; float* h; float* gates;            /* (B,Q), (B,Q) raw h, (B,4Q) activated gates */
    const float* lnq_g; const float* lnq_b;      /* query_lstm_layernorm */
    float* qcur; float* st_q;                    /* (B,Q) dropout(LN(h)), (B,2) */
    float p_q; uint32_t site_q;
    /* attention streams */
    const float* Kp[2]; const float* Vp[2];      /* (B,P,Q), (B,P,H) */
    const float* lnc_g[2]; const float* lnc_b[2];/* output_layer.2 of each AttentionShare */
    float* cpre[2]; float* ctx[2]; float* st_c[2];   /* (B,H) pre-tanh context, (B,H) output, (B,2) */
    float* alpha;                                /* (B, nstream*P) */
    float p_att[2]; uint32_t site_att[2];
    int32_t B, Q, H, P, nstream;
    float scale, eps;
    int32_t kv_div;                              /* <= 1: row b attends over K'[b], V'[b].  k > 1 (beam search: the k beams of a
                                                    clip are consecutive rows): row b attends over K'[b / k], V'[b / k], which
                                                    then hold one block per CLIP -- the beams share the lines in L2 instead of
                                                    each streaming its own expanded copy */
    uint64_t seed; const uint64_t* seed_ptr;
} dlsg_dec_mid_args;
int dlsg_dec_mid_fwd(const dlsg_dec_mid_args* a, void* stream);
/* language LSTM cell pointwise (+dropout on h) -> tanh(lang_lstm_layernorm(h)) for the vocab projection. */
typedef struct {
    const float* slabs; int32_t nslab; int32_t pad_; int64_t slab_stride;     /* (S,B,4D) */
    const float* b_ih; const float* b_hh;
    const float* c_prev; float* c; float* hd; float* gates;    /* (B,D) ..., hd = dropout(h) (next state), (B,4D) */
    const float* ln_g; const float* ln_b;
    float* dout; float* st_l;                   /* (B,D) tanh(LN(hd)), (B,2) */
    float p; uint32_t site;
    int32_t B, D;
    float eps;
    uint64_t seed; const uint64_t* seed_ptr;
    /* optional (s_coins != NULL): scheduled sampling of the NEXT step's word inside this launch (models/layer.py:432-441).  When
     * s_coins[s_t] == 0 every workgroup projects its row onto the vocabulary (s_W (V x D), s_b), takes the first maximum, writes
     * it to s_ids[b] and its embedding row (s_E (V x W), word dropout s_p / s_site at row s_row0 + b, the mask stream of
     * dlsg_embed_fwd / dlsg_select_embed) to s_we; when the coin is set (teacher-forced step, ids and embedding filled in up
     * front) nothing more happens.  One weight read per workgroup: for vocabularies up to ~2 M weights; larger ones keep
     * dlsg_gemm + dlsg_select_embed.  Replaces two launches per word step that do nothing on ~19 steps out of 20. */
    const int32_t* s_coins; int32_t s_t; int32_t s_V;
    const float* s_W; const float* s_b;
    const float* s_E; int32_t s_Wd; uint32_t s_site;
    int64_t* s_ids; float* s_we; int64_t s_ldwe;
    int64_t s_row0; float s_p; float pad2_;
} dlsg_dec_tail_args;
int dlsg_dec_tail_fwd(const dlsg_dec_tail_args* a, void* stream);
/* Backward of dlsg_dec_mid_fwd for one word step, one workgroup per batch row.  Consumes the slabs of the language
 * cell's input-gradient GEMM directly (their sum is [d ctx_0 | d ctx_1 | d q_cur | d lang_h(recurrent)]), runs the
 * output_layer LayerNorm backward of each stream, the attention backward (score / softmax / context), the
 * query_lstm_layernorm backward and the query cell backward.  The gradients of the cached K', V' are NOT accumulated
 * here: the step leaves d(pre-tanh context) and d(score) in per-step buffers and dlsg_decatt_cache_grads contracts them
 * over the word loop afterwards (no read-modify-write of the (B,P,Q)+(B,P,H) caches per word).
 * Replaces dlsg_slab_reduce + 2 x dlsg_rowln_bwd + dlsg_decatt_bwd + dlsg_rowln_bwd + dlsg_lstm_pw_bwd. */
typedef struct {
    const float* slabs; int32_t nslab; int32_t write_rec; int64_t slab_stride;   /* (S,B,ns*H+Q+D), rows dense */
    float* dlh_rec;                      /* (B,D) out when write_rec: sum of the last D columns */
    const float* cpre[2]; const float* st_c[2]; const float* lnc_g[2];
    float* part_c[2];                    /* (B,2,H) per-row dgamma | dbeta of each output_layer LayerNorm */
    float* dcpre[2];                     /* (B,H) out */
    float p_att[2]; uint32_t site_att[2];
    const float* Kp[2]; const float* Vp[2];
    const float* alpha; const float* dalpha;     /* (B,ns*P); dalpha optional */
    float* ds;                           /* (B,ns*P) out */
    const float* qh; const float* st_q; const float* lnq_g;
    float* part_q;                       /* (B,2,Q) */
    float p_q; uint32_t site_q;
    const float* rec_slabs; int32_t rec_nslab; int32_t pad_; int64_t rec_slab_stride; int64_t rec_ld;
                                         /* (S',B,>=Q): recurrent d h_query from step t+1, NULL at the last step */
    const float* gates; const float* c; const float* c_prev;   /* (B,4Q), (B,Q), (B,Q) */
    float* dc;                           /* (B,Q) in/out: cell-state gradient */
    float* dgates;                       /* (B,4Q) out */
    int32_t B, Q, H, D, P, nstream;
    float scale; float pad2_;
    uint64_t seed; const uint64_t* seed_ptr;
} dlsg_dec_mid_bwd_args;
int dlsg_dec_mid_bwd(const dlsg_dec_mid_bwd_args* a, void* stream);
/* dK'[s][b,p,:] = sum_t ds[t,b,s*P+p] * q_cur[t,b,:],  dV'[s][b,p,:] = sum_t alpha[t,b,s*P+p] * dcpre[s][t,b,:]
 * (time-major (L,B,.) inputs; outputs written, not accumulated). */
typedef struct {
    const float* alpha; const float* ds;         /* (L,B,ns*P) */
    const float* qcur;                           /* (L,B,Q) */
    const float* dcpre[2];                       /* (L,B,H) */
    float* dKp[2]; float* dVp[2];                /* (B,P,Q), (B,P,H) */
    int32_t L, B, Q, H, P, nstream;
} dlsg_decatt_cache_grads_args;
int dlsg_decatt_cache_grads(const dlsg_decatt_cache_grads_args* a, void* stream);

/* ---------------------------------------------------------------- LSTM cell pointwise (nn.LSTM layer.py:52, nn.LSTMCell :571,593)
 * gates = sum_s slabs[s] (+ addend) (+ b_ih + b_hh), PyTorch gate order i,f,g,o;
 * c = f*c_prev + i*g; h = o*tanh(c).  Saves activated gates (B,4H) for backward.  h is written to up to two
 * destinations (strided); h2 receives dropout(h) when p>0 (lang_lstm_drop layer.py:594). */
typedef struct {
    const float* slabs; int32_t nslab; int32_t pad_; int64_t slab_stride;
    const float* addend; int64_t ldadd;        /* optional precomputed gate part (B,4H) */
    const float* b_ih; const float* b_hh;      /* optional */
    const float* c_prev; int64_t ldcp;         /* NULL = zeros */
    float* c; int64_t ldc_;
    float* h; int64_t ldh;                     /* raw h (recurrent state) */
    float* h2; int64_t ldh2;                   /* optional second copy (dropout applied if p>0) */
    float* gates; int64_t ldg;                 /* (B,4H) activated gates (row stride ldg), optional */
    int32_t B, H;
    float p; uint32_t site; uint64_t seed;
    const uint64_t* seed_ptr;
} dlsg_lstm_pw_args;
int dlsg_lstm_pw_fwd(const dlsg_lstm_pw_args* a, void* stream);
/* `count` (1 or 2) descriptors of equal B, H in one launch: the two directions of a BiLSTM step */
int dlsg_lstm_pw_fwd_n(const dlsg_lstm_pw_args* a, int count, void* stream);
/* backward: dh (B,H) [+ dh2 through dropout], dc_next -> dgates (B,4H pre-activation grads), dc_prev */
typedef struct {
    const float* gates; int64_t ldg;           /* activated gates from forward, row stride ldg */
    const float* c; int64_t ldc_;
    const float* c_prev; int64_t ldcp;
    const float* dh; int64_t lddh;             /* optional */
    const float* dh2; int64_t lddh2;           /* optional, goes through the dropout mask */
    const float* dh3; int64_t lddh3;           /* optional, summed with dh2 before the mask */
    const float* dh4; int64_t lddh4;           /* optional, summed with dh2 before the mask */
    int32_t dh4_nslab; int32_t pad4_; int64_t dh4_slab_stride;   /* dh4_nslab > 1: dh4 is a slab stack, summed on the fly */
    const float* dc_next; int64_t lddcn;       /* optional */
    float* dgates; int64_t lddg;               /* (B,4H) pre-activation gate grads, row stride lddg */
    float* dc_prev; int64_t lddcp;
    int32_t B, H;
    float p; uint32_t site; uint64_t seed;
    const uint64_t* seed_ptr;
} dlsg_lstm_pw_bwd_args;
int dlsg_lstm_pw_bwd(const dlsg_lstm_pw_bwd_args* a, void* stream);
int dlsg_lstm_pw_bwd_n(const dlsg_lstm_pw_bwd_args* a, int count, void* stream);

/* ---------------------------------------------------------------- small data movement on the path
 * mean over P proposals (layer.py:407-410): out[b, off + h] = mean_p x[b,p,h]; and its backward (accumulating) */
int dlsg_mean_rows_fwd(const float* x, float* out, int64_t ldo, int B, int P, int H, void* stream);
int dlsg_mean_rows_bwd(const float* dout, int64_t lddo, float* dx, int B, int P, int H, int accum, void* stream);
/* embedding gather + dropout (layer.py:421-422,438-439): out[r, :] = drop(E[ids[r], :]); the dropout mask of
 * element (r, j) is keyed by (row0 + r) * W + j so a slice of a larger call reproduces the same mask. */
int dlsg_embed_fwd(const float* E, const int64_t* ids, float* out, int64_t ldo, int rows, int W, float p, uint64_t seed,
                   uint32_t site, int64_t row0, const uint64_t* seed_ptr, void* stream);
/* dE[ids[r], :] += drop(dout[r, :])   (atomic adds; rows sharing an id collide) */
int dlsg_embed_bwd(const float* dout, int64_t lddo, const int64_t* ids, float* dE, int rows, int W, float p,
                   uint64_t seed, uint32_t site, int64_t row0, const uint64_t* seed_ptr, void* stream);
/* scheduled sampling on device (layer.py:432-439): id[b] = coins[t] ? captions[b*L + t] : argmax(logits[b, :]);
 * ids_out[b] = id; out[b, :] = drop(E[id, :]).  coins is a device int32 array, so a captured graph is invariant to the
 * coin pattern.  prefilled != 0: the caller wrote the teacher-forced choice of this step into ids_out / out up front (one
 * dlsg_embed_fwd over all steps), so the launch is a no-op when coins[t] is set. */
int dlsg_select_embed(const float* logits, int64_t ld, int V, const int64_t* captions, int L, int t, const int32_t* coins,
                      const float* E, int64_t* ids_out, float* out, int64_t ldo, int rows, int W, float p, uint64_t seed,
                      uint32_t site, int64_t row0, const uint64_t* seed_ptr, int prefilled, void* stream);
/* argmax over logits rows (first max wins, like torch.max) */
int dlsg_argmax(const float* logits, int64_t ld, int64_t* ids, int rows, int V, void* stream);
/* strided 2-d copy / add: dst[r*ldd + j] (+)= src[r*lds + j] */
int dlsg_copy2d(const float* src, int64_t lds, float* dst, int64_t ldd, int rows, int n, int accum, void* stream);
/* elementwise dropout with the stateless mask: y = x * keep(seed, site, r*n+j)/(1-p) */
int dlsg_dropout(const float* x, int64_t ldx, float* y, int64_t ldy, int rows, int n, float p, uint64_t seed,
                 uint32_t site, const uint64_t* seed_ptr, void* stream);
int dlsg_fill(float* dst, int64_t n, float value, void* stream);
/* dst[r, :] = src[idx[r], :]  (beam-search state reorder by back-pointer, allennlp_beamsearch.py:248-260) */
int dlsg_gather_rows(const float* src, int64_t lds, const int64_t* idx, float* dst, int64_t ldd, int rows, int n, void* stream);
/* dst[b,t,:] = src[t,b,:]: time-major decoder buffers -> the (B,L,V) layout Decoder.forward returns (layer.py:447) */
int dlsg_permute_tb(const float* src, float* dst, int T, int B, int n, void* stream);

/* ---------------------------------------------------------------- DiscV2 critic: the blocks of its schedule
 * `DiscV2.forward` (models/model.py:143-166), `PSLScore2.forward` (models/layer.py:690-715), `SelfAttention` with the caption mask
 * (models/sublayer.py:63-82), `LatentPSL` (sublayer.py:189-198), `ResBlock` (sublayer.py:107-119) and the WGAN-GP critic update
 * around them (run_gun.py:339-381: three critic forwards, gradient penalty with create_graph=True, loss backward) as explicit
 * passes of dlsg_amd/critic.py.  Every block exists at three levels:
 *   fwd   the block's forward;
 *   bwd   its vector-Jacobian product (inputs; parameters where it has any);
 *   bwd2  the DIRECTIONAL DERIVATIVE of (fwd, bwd) along a tangent U of the block's input at fixed output cotangent -- what the
 *         gradient penalty's second-order term needs of a block: the derivative of fwd is the tangent handed to the next block,
 *         the derivative of bwd's input gradient an extra cotangent on the block's input, the derivative of bwd's parameter
 *         gradient an extra parameter gradient.
 * Activations are batch-major: (captions, L words, 512) dense; `captions` = n blocks of B ("caption sets" scored against the same
 * B clips: caption i belongs to clip i % B).  acc_lo / acc_hi: a caption (or row) range whose outputs are ADDED to what the
 * buffer holds (the extra cotangents of the bwd2 pass) instead of written.  Dropout is the stateless mask of the generator path
 * (keyed by seed + *seed_ptr, site, element index); p = 0 switches it off.  L <= 32, T <= 8 proposals, width 512. */
#define DLSG_CRIT_C 512
#define DLSG_CRIT_LMAX 32
#define DLSG_CRIT_TMAX 8

/* Vocabulary projection glue (models/model.py:143-144 on one-hot / logit inputs, run_gun.py:447-451,358-360).
 * embed_mix: proj_tm (L,B,512) = logits W^T without bias; ids (B,L) or NULL; W (512,V); eps (B) or NULL.
 *   ng = 3: h[0] = W[:, ids] + bias (real captions as a gather of weight columns), h[1] = proj + bias, h[2] = eps h[0] + (1-eps) h[1];
 *   ng = 1: h[0] = proj + bias.   h (ng, B, L, 512).
 * embed_mix_bwd: ch (ng,B,L,512) -> dhr (B,L,512) = ch[0] + eps ch[2] and dhf_tm (L,B,512) = ch[1] + (1-eps) ch[2]  (ng = 1: dhf_tm = ch[0]^T).
 * vocab_scatter: dW[:, ids[b,l]] += dhr[b,l,:] in a fixed order (the first row carrying an id adds all rows of that id). */
int dlsg_crit_embed_mix(const float* proj_tm, const int64_t* ids, const float* W, const float* bias, const float* eps, float* h,
                        int ng, int B, int L, int V, void* stream);
int dlsg_crit_embed_mix_bwd(const float* ch, const float* eps, float* dhr, float* dhf_tm, int ng, int B, int L, void* stream);
int dlsg_crit_vocab_scatter(const float* dhr, const int64_t* ids, float* dW, int rows, int V, void* stream);

/* ResBlock head (sublayer.py:110-119; its ReLU is in place, so the skip carries relu(x)): z = x * [ref > 0];
 * y = z (+ bias_scale * bias); taps[i, l, 3 c + k] = z[i, l + k - 1, c] (zero outside the caption): the three shifted copies that
 * turn Conv1d(512, 512, 3, padding = 1) into ONE product with conv.weight viewed as (512, 1536).  x, ref, y (n, L, 512), taps (n, L, 1536).
 * bwd: dx = (dy + sum_k dtaps[i, l - k + 1, 3 c + k]) * [ref > 0]. */
int dlsg_crit_relu_taps(const float* x, const float* ref, const float* bias, float bias_scale, float* y, float* taps, int n, int L,
                        void* stream);
int dlsg_crit_relu_taps_bwd(const float* dy, const float* dtaps, const float* ref, float* dx, int n, int L, void* stream);

/* (tanh +) LayerNorm with dropouts: z = drop_pre(x); t = pre_tanh ? tanh(z) : z; y = drop_post(LN(t) gamma + beta), rows of N = 64..1024
 * columns (N % 64 == 0), `groups` <= 4 same-shape blocks of `rows` rows with their own arrays (the critic's two proposal heads side
 * by side); block g uses dropout sites site_* + g; the mask of element (r, j) is keyed by (row0 + r) * N + j.
 *   bwd : dy = sum of ndy <= 3 arrays; dx (rows [acc_lo, acc_hi) of every block added to), dgamma / dbeta (NULL: not wanted;
 *         extra[g] (2, N), optional, is added to (dgamma, dbeta)); ws = dlsg_cln_ws_floats(rows, N) * groups floats of scratch
 *   bwd2: U = tangent of x -> gdy = tangent of y, gx = derivative of dx, gpart[g] (2, N) = derivative of (dgamma, dbeta) */
#define DLSG_CLN_MAXG 4
typedef struct {
    const float* x[DLSG_CLN_MAXG]; const float* gamma[DLSG_CLN_MAXG]; const float* beta[DLSG_CLN_MAXG];
    float* y[DLSG_CLN_MAXG];
    const float* dy[3][DLSG_CLN_MAXG]; float* dx[DLSG_CLN_MAXG];
    float* dgamma[DLSG_CLN_MAXG]; float* dbeta[DLSG_CLN_MAXG]; const float* extra[DLSG_CLN_MAXG];
    const float* U[DLSG_CLN_MAXG]; float* gx[DLSG_CLN_MAXG]; float* gdy[DLSG_CLN_MAXG]; float* gpart[DLSG_CLN_MAXG];
    float* ws;
    int32_t rows, N, groups, pre_tanh, ndy, acc_lo, acc_hi;
    int32_t defer; /* != 0: dlsg_cln_bwd / dlsg_cln_bwd2 leave the per-workgroup partial sums in ws -- [group][dgamma | dbeta]
                    * [dlsg_cln_ws_floats(rows, N) / (2 N) rows][N], bwd2 fills the dgamma half only (its dbeta is zero) -- and do
                    * not fold them: the caller sums the rows with the other column sums of its step (dlsg_crit_colsum) */
    float eps, p_pre, p_post; uint32_t site_pre, site_post, pad2_;
    uint64_t seed; const uint64_t* seed_ptr; int64_t row0;
} dlsg_cln_args;
int64_t dlsg_cln_ws_floats(int rows, int N);
int dlsg_cln_fwd(const dlsg_cln_args* a, void* stream);
int dlsg_cln_bwd(const dlsg_cln_args* a, void* stream);
int dlsg_cln_bwd2(const dlsg_cln_args* a, void* stream);

/* Masked self-attention core of one caption on rows [K | Q | V] (n, L, 1536) (sublayer.py:66-78 with att_mask[b,i,j] =
 * smask[b,i] smask[b,j], run_gun.py:164-166; smask (B, L), caption i uses smask[i % B]):
 *   w = softmax_j(scale K_i . Q_j, masked entries -9e15), ctx = w V.     bwd: dctx -> dKQV.
 *   bwd2: U (n, L, 1536) tangent of [K | Q | V], dctx fixed, w as the forward saved it -> Uctx (tangent of ctx), gKQV (derivative of
 *         dKQV; w varies with K, Q). */
typedef struct {
    const float* KQV; const float* smask; float* w; float* ctx;
    const float* dctx; float* dKQV;
    const float* U; float* Uctx; float* gKQV;
    int32_t n, B, L, acc_lo, acc_hi, pad_;
    float scale, pad2_;
} dlsg_crit_sa_args;
int dlsg_crit_sa_fwd(const dlsg_crit_sa_args* a, void* stream);
int dlsg_crit_sa_bwd(const dlsg_crit_sa_args* a, void* stream);
int dlsg_crit_sa_bwd2(const dlsg_crit_sa_args* a, void* stream);

/* PSLScore2's word -> proposal graph (layer.py:697-707), both heads per launch: a[h] (n, L, 512) words, e[h] (B, T, 512) proposals
 * of the clips (caption i scores clip i % B):
 *   P = softmax over the words of scale a e^T (n, L, T); adj = P * smask (mask AFTER the softmax, :703-704); wgt = sum_l adj (n, T);
 *   aggpre = adj^T a (n, T, 512).
 *   bwd : d_agg (n, T, 512), d_wgt (n, T) -> da (n, L, 512), de (n, T, 512) per caption (NULL: not wanted; the caller sums a clip's captions)
 *   bwd2: Ua = tangent of a, P as the forward saved it -> Uagg, Uwgt (tangents), ga, ge (derivatives of da, de) */
typedef struct {
    const float* a[2]; const float* e[2]; const float* smask;
    float* P[2]; float* wgt[2]; float* aggpre[2];
    const float* d_agg[2]; const float* d_wgt[2]; float* da[2]; float* de[2];
    const float* Ua[2]; float* Uagg[2]; float* Uwgt[2]; float* ga[2]; float* ge[2];
    int32_t n, B, L, T, acc_lo, acc_hi;
    float scale, pad_;
} dlsg_crit_pattn_args;
int dlsg_crit_pattn_fwd(const dlsg_crit_pattn_args* a, void* stream);
int dlsg_crit_pattn_bwd(const dlsg_crit_pattn_args* a, void* stream);
int dlsg_crit_pattn_bwd2(const dlsg_crit_pattn_args* a, void* stream);

/* Text summary (LatentPSL(512, 1), sublayer.py:189-198, no mask) and the fusion weights (models/model.py:163-165):
 *   adj = softmax_l(words theta) (n, L); u = adj^T words (n, 512); sent = drop(LN(tanh(u))); fus = softmax(sent fusion^T) (n, 2)
 *   bwd : d_fus (n, 2) -> dwords (n, L, 512), part (n, 5, 512) per-caption partials of [dtheta, dgamma, dbeta, dfusion_0, dfusion_1]
 *         (NULL: not wanted)
 *   bwd2: U (n, L, 512) tangent of words -> Ufus, gwords, gpart (n, 5, 512) */
typedef struct {
    const float* words; const float* theta; const float* gamma; const float* beta; const float* fusion;
    float* adj; float* u; float* sent; float* fus;
    const float* d_fus; float* dwords; float* part;
    const float* U; float* Ufus; float* gwords; float* gpart;
    int32_t n, L, acc_lo, acc_hi;
    float eps, p; uint32_t site, pad_;
    uint64_t seed; const uint64_t* seed_ptr; int64_t row0;
} dlsg_crit_tsum_args;
int dlsg_crit_tsum_fwd(const dlsg_crit_tsum_args* a, void* stream);
int dlsg_crit_tsum_bwd(const dlsg_crit_tsum_args* a, void* stream);
int dlsg_crit_tsum_bwd2(const dlsg_crit_tsum_args* a, void* stream);

/* Pair scores, batch means, critic output (sublayer.py:304-306, layer.py:709-714, models/model.py:165-166), heads h = 0, 1:
 *   pair[h][i,t] = sum_c v[h][i % B,t,c] s[h][i,t,c] wc[h][c] + bc[h];  score[h][i] = sum_t pair wgt / sum_t wgt;
 *   both[g][h] = mean over the B captions of set g of score[h] (PSLScore2 ends with a mean over ITS batch);  out[i] = sum_h both[g(i)][h] fus[i][h]
 *   v = tanh(v_pre) (B, T, 512), s = tanh(s_pre) (n, T, 512): the backward returns the gradients of the PRE-activations.
 *   bwd : d_out (n) -> d_fus (n, 2), c_spre[h] (n, T, 512), c_vpre[h] (n, T, 512) per caption (NULL: not wanted), d_wgt[h] (n, T),
 *         part_wc[h] (n, 512) per-caption partials of classify.weight's gradient and dbc (2) (NULL: not wanted); scratch (4 ng + 8) floats
 *   bwd2: ONE caption set (n == B); tangents Uspre[h] (n, T, 512) of s_pre, Uwgt[h] (n, T), Ufus (n, 2) -> the derivatives of bwd's
 *         outputs: g_fus, g_spre, g_vpre, g_wgt, gpart_wc, g_dbc; scratch (2 n + 16) floats */
typedef struct {
    const float* v[2]; const float* s[2]; const float* wc[2]; const float* bc[2]; const float* wgt[2]; const float* fus;
    float* pair[2]; float* score[2]; float* both; float* out;
    const float* d_out; float* d_fus; float* c_spre[2]; float* c_vpre[2]; float* d_wgt[2]; float* part_wc[2]; float* dbc;
    const float* Uspre[2]; const float* Uwgt[2]; const float* Ufus;
    float* scratch;
    int32_t n, B, T, ng, acc_lo, acc_hi;
} dlsg_crit_score_args;
int dlsg_crit_score_fwd(const dlsg_crit_score_args* a, void* stream);
int dlsg_crit_score_bwd(const dlsg_crit_score_args* a, void* stream);
int dlsg_crit_score_bwd2(const dlsg_crit_score_args* a, void* stream);

/* Gradient penalty and the update's loss values (run_gun.py:362-375) from g = d(sum mixed scores)/d(mixed projection) (B, L, 512) and
 * gG = g (W W^T): q_b = sum g gG = |d mixed score_b / d mixed caption_b|^2, gn = sqrt(max(q, 1e-24)), penalty = mean (gn - 1)^2,
 * c_b = d penalty / d q_b (0 where q was clamped).  out (3B): scores of [real | fake | mixed].
 * stats[0..5) = loss_D = mean fake - mean real + 10 penalty, mean real, mean fake, penalty, mean real - mean fake;
 * vseed = 20 c_b gG (= 10 d penalty / d g), gsc = 10 c_b g.  q: B floats of scratch. */
int dlsg_crit_gp(const float* g, const float* gG, const float* out, float* stats, float* vseed, float* gsc, float* q, int B, int L,
                 void* stream);

/* Top-k proposals of a clip by attention mass (layer.py:694-696): head 0 = the first P columns of alpha (B, L, 2P) (row stride lda,
 * clip stride sa), head 1 = the last P; idx (2, B, T) int64 = rows (h B + b) P + p of the (2 B P, .) proposal embeddings, by
 * decreasing sum_l alpha smask (ties: the lower p).  unselect: dst (R, n) row idx[r] = src[r], every other row zero. */
int dlsg_crit_topk(const float* alpha, int64_t sa, int64_t lda, int na, const float* smask, int64_t* idx, int B, int L, int P, int T,
                   void* stream);
int dlsg_crit_unselect(const float* src, const int64_t* idx, float* dst, int rows_src, int rows_dst, int per, int n, void* stream);

/* `count` <= DLSG_CRIT_REDUCE_MAX sums of slabs in one launch: out[i] = scale * sum_{k < nslab} src[k * stride + i], i < n (n, stride
 * multiples of 4, 16-byte aligned): the K-split partial weight gradients of a critic update, folded in slab order. */
#define DLSG_CRIT_REDUCE_MAX 16
typedef struct {
    const float* src; int64_t stride; int64_t n; int32_t nslab; float scale;
    float* out;
} dlsg_crit_reduce_desc;
int dlsg_crit_reduce(const dlsg_crit_reduce_desc* d, int count, void* stream);

/* `count` <= DLSG_CRIT_COLSUM_MAX column sums in one launch: out[j] = scale * (sum_r part[r, j] + sum_r part_b[r, j]) (part_b optional), also written
 * to out_b when set (bias_ih / bias_hh of an LSTM receive the same gradient).  Fixed order of additions.  n <= 2048. */
#define DLSG_CRIT_COLSUM_MAX 48
typedef struct {
    const float* part; int64_t ld; int32_t rows, n;
    const float* part_b; int64_t ld_b; int32_t rows_b, pad_;
    float* out; float* out_b;
    float scale, pad2_;
} dlsg_crit_colsum_desc;
int dlsg_crit_colsum(const dlsg_crit_colsum_desc* d, int count, void* stream);

/* ---------------------------------------------------------------- loss + optimizer (run_gun.py:189-198, :91)
 * Ragged CrossEntropy: row (b,t) counts iff t < lens[b]; loss = mean over counted rows; dlogits written for all
 * rows (zeros for padded ones).  row_loss (B*L) is scratch; loss[0] receives the mean.  time_major: logits and
 * dlogits are laid out (L,B,V) (the decoder's internal layout) instead of (B,L,V); targets stay (B,L). */
int dlsg_ce_ragged(const float* logits, const int64_t* targets, const int64_t* lens, float* dlogits, float* row_loss,
                   float* loss, int B, int L, int V, int time_major, void* stream);
int dlsg_log_softmax(const float* logits, float* out, int rows, int V, void* stream);
/* One beam-search step for every batch item (BeamSearch.search, allennlp_beamsearch.py:140-260, per-node k == k):
 * log-softmax + per-beam top-k over the vocabulary + top-k over the k*k summed candidates, in one launch.
 * logits (B*k, V) rows (ld floats apart); last (B*k) the tokens fed to this step (a beam whose last token is `end`
 * offers only `end` at log-prob 0); last_lp (B*k) running log-probs.  first != 0: step 0 (only beam 0 of each group).
 * Out: pred (B*k) int64 chosen classes, new_lp (B*k), back (B*k) int64 parent beam, rows (B*k) int64 = b*k + back (the
 * gather indices for the recurrent state), ended_count += number of chosen `end` tokens (optional device counter). */
typedef struct {
    const float* logits; int64_t ld;
    const int64_t* last; const float* last_lp;
    int64_t* pred; float* new_lp; int64_t* back; int64_t* rows;
    int32_t* ended_count;
    int32_t B, k, V, end, first, pad_;
} dlsg_beam_select_args;
int dlsg_beam_select(const dlsg_beam_select_args* a, void* stream);
/* dst_i[r,:] = src_i[rows[r],:] (dense rows of n[i] floats) for count <= 4 arrays in one launch */
typedef struct {
    const float* src[4]; float* dst[4]; int32_t n[4];
    const int64_t* rows; int32_t nrows, count;
} dlsg_gather_multi_args;
int dlsg_gather_rows_multi(const dlsg_gather_multi_args* a, void* stream);
/* torch.optim.Adam semantics (no weight decay, no amsgrad); step = 1-based step count; grad_scale folds 1/world */
int dlsg_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, int step,
              float grad_scale, const float* hyper, const int32_t* guard, void* stream);
/* hyper (optional, device): {lr / (1 - b1^step), sqrt(1 - b2^step)} -- overrides lr/step so a captured graph can be
 * replayed with a new step count.  guard (optional, device): the launch updates nothing when *guard != 0 -- the `err` word of the
 * persistent recurrent kernels (dlsg_bilstm_*, dlsg_lstm_seq): a step whose hand-off timed out has invalid gradients. */


/* ---------------------------------------------------------------- persistent BiLSTM recurrence (csrc/bilstm.hip)
 * Replaces the time loop of `nn.LSTM(H, H, bidirectional=True, batch_first=True)` in EncoderVisual (models/layer.py:26,52)
 * once the input half of the gates xg = e W_ih^T is known for all steps: ONE launch runs all T steps of both directions,
 * 2 * H/8 workgroups (one per CU, all resident), each keeping its 32 x H slice of W_hh in LDS and its cell states in
 * registers; h_t travels between workgroups through `hx` with write-through stores and per-step flags.
 * Layouts: xg[d] (B*T, >= 4H) rows b*T + t, gate order i,f,g,o (PyTorch); out (B, T, 2H) with direction d in columns
 * [dH, (d+1)H); hprev[d] (B, T, H) = h of the PREVIOUS step of direction d at each time index (the caller zero-fills it: the
 * first step of a direction is never written); c[d] (B, T, H); gates[d] (B, T, 4H) activated gates (what the backward reads).
 * Scratch owned by the caller: hx = dlsg_bilstm_hx_floats(T, H) floats, flags = dlsg_bilstm_flag_words(T, H) 32-bit words
 * (zeroed by the call itself), err = optional int32 that is set to 1 if a workgroup timed out waiting for another (the result
 * is then invalid; the device is never left spinning).  dlsg_bilstm_supported: B <= 64, H in {64, 512, 1024} and the current
 * device has at least 2 * H/8 compute units (all workgroups must be resident together). */
typedef struct {
    const float* xg[2];
    int64_t ldxg;
    const float* w_hh[2];
    const float* b_ih[2];
    const float* b_hh[2];
    float* out;
    float* hprev[2];
    float* c[2];
    float* gates[2];
    float* hx;
    uint32_t* flags;
    int32_t* err;
    int32_t B, T, H, pad_;
} dlsg_bilstm_args;
int dlsg_bilstm_supported(int B, int T, int H);
int64_t dlsg_bilstm_hx_floats(int T, int H);
int64_t dlsg_bilstm_flag_words(int T, int H);
int dlsg_bilstm_fwd(const dlsg_bilstm_args* a, void* stream);
/* Backward through time of the same recurrence, again ONE launch: from the saved activated gates / cell states and
 * dout (B, T, 2H) = d loss / d out, writes dgates[d] (B, T, 4H) = d loss / d (gate pre-activations) for every step (what the
 * weight / input gradient products after the loop contract).  Scratch: gx, px = dlsg_bilstm_bwd_x_floats(T, H) floats each
 * (per-step exchange of dG and of the K-quarter partials of dh), flags = 2 * dlsg_bilstm_flag_words(T, H) words. */
typedef struct {
    const float* gates[2];
    const float* c[2];
    const float* dout;
    const float* w_hh[2];
    float* dgates[2];
    float* gx;
    float* px;
    uint32_t* flags;
    int32_t* err;
    int32_t B, T, H, pad_;
} dlsg_bilstm_bwd_args;
int64_t dlsg_bilstm_bwd_x_floats(int T, int H);
int dlsg_bilstm_bwd(const dlsg_bilstm_bwd_args* a, void* stream);

/* ---------------------------------------------------------------- DiscV2's LSTM, a whole sequence per launch
 * Replaces `self.lstm = nn.LSTM(512, 512, batch_first=True)` of DiscV2 (models/model.py:122,139) inside a WGAN-GP critic
 * update (run_gun.py:352-371), where autograd differentiates it twice: level 0 = the forward recurrence, level 1 = its
 * backward through time (with the extra gradient inputs the gradient penalty's graph feeds in), level 2 = the backward of
 * that backward.  One persistent launch per level for all L steps (csrc/critic_lstm.hip; the per-step forms are
 * cell formulas in that file's header).  Tensors are contiguous, time-major (L, n, .) or batch-major (n, L, .) (`batch_major`):
 * gate tensors 4H wide (gate order i, f, g, o), states H wide.  n <= 256, H in {64, 512}; W = weight_hh (4H, H).
 *   level 0: addend = x W_ih^T + b_ih + b_hh  ->  As (pre-activations), Hs, Cs
 *   level 1: As, Cs, dHs, optional dAs / dCs  ->  DA (d As, injections included), DH, DC (total gradients reaching h_t, c_t)
 *   level 2: As, Cs, DH, DC, addend = gradient w.r.t. DA  ->  addend_out = Ubar (gradient w.r.t. dAs; optional, may alias addend),
 *            gA, gC (its last step is zero), gDH (gradient w.r.t. dHs), gDC (w.r.t. dCs)
 * xbuf / xbuf2: dlsg_lstm_seq_x_floats floats each (xbuf2 only at level 1), 16-byte aligned; flags: dlsg_lstm_seq_flag_words
 * words, zeroed by the call; err (optional): set non-zero if a workgroup timed out waiting (result then invalid). */
typedef struct {
    const float* addend;
    float* addend_out;
    const float* W;
    float* As; float* Hs; float* Cs;                 /* level 0 writes, levels 1-2 read As, Cs */
    const float* dHs; const float* dAs; const float* dCs;
    float* DA; float* DH; float* DC;                 /* level 1 writes, level 2 reads DH, DC */
    float* gA; float* gC; float* gDH; float* gDC;
    float* xbuf; float* xbuf2;
    uint32_t* flags;
    int32_t* err;
    int32_t L, n, H;
    int32_t batch_major;                             /* 0: (L, n, .) arrays; 1: (n, L, .) -- a sequence's steps are consecutive rows */
    const float* b_ih; const float* b_hh;            /* level 0, optional (both or neither): added to the addend */
    float* Hprev;                                    /* level 0, optional: Hprev_t = h_{t-1} (zeros at t = 0), layout of Hs */
    float* gDHprev;                                  /* level 2, optional: gDHprev_t = gDH_{t-1} */
} dlsg_lstm_seq_args;
int dlsg_lstm_seq_supported(int L, int n, int H);
int64_t dlsg_lstm_seq_x_floats(int L, int n, int H);
int64_t dlsg_lstm_seq_flag_words(int L, int n, int H);
int dlsg_lstm_seq(const dlsg_lstm_seq_args* a, int level, void* stream);

/* ---------------------------------------------------------------- gradient all-reduce over RCCL / xGMI
 * Replaces the gradient exchange of `DistributedDataParallel(model, find_unused_parameters=True)` over NCCL
 * (run_gun.py:63-64, train_debug.py:20): one process per GPU, sum of ranges of the flat gradient arena over all ranks,
 * in place; the mean (1/world) is folded into dlsg_adam's grad_scale.  librccl is resolved at run time (the instance a
 * PyTorch process has already mapped is reused), so the library loads without it; these calls then return DLSG_ENOCOMM.
 *
 * The communicator handle is the ONLY state the library keeps and the caller owns it:
 *   rank 0: dlsg_comm_unique_id(id)  ->  the caller ships the 128 bytes to every rank (any side channel: torch.distributed
 *   store, MPI, a file)  ->  every rank, with its device current: dlsg_comm_init(&c, id, world, rank) (collective)
 *   ...  dlsg_allreduce_bucket(c, grads + lo, hi - lo, stream) per gradient bucket, same order on every rank  ...
 *   dlsg_comm_destroy(c).
 * The collectives are asynchronous on `stream` and may be stream-captured: dlsg_amd.Trainer forks a side stream inside the
 * capture of the train step, so the bucket all-reduces overlap the remaining backward within ONE replayed hipGraph. */
#define DLSG_COMM_ID_BYTES 128
typedef struct dlsg_comm dlsg_comm;
int dlsg_comm_unique_id(void* id128);
int dlsg_comm_init(dlsg_comm** out, const void* id128, int world, int rank);
int dlsg_comm_destroy(dlsg_comm* c);
int dlsg_comm_info(const dlsg_comm* c, int32_t* world, int32_t* rank, int32_t* rccl_version);
int dlsg_allreduce_bucket(dlsg_comm* c, float* grads, int64_t count, void* stream);
/* n ranges of one bucket as one RCCL group (one fused launch) */
int dlsg_allreduce_buckets(dlsg_comm* c, float* const* grads, const int64_t* counts, int n, void* stream);
/* words[i] <- max over the ranks (int32), on `stream`: the persistent kernels' time-out word travels with the first gradient
 * bucket, so that dlsg_adam's guard skips the update on EVERY rank or on none. */
int dlsg_allreduce_max_i32(dlsg_comm* c, int32_t* words, int64_t count, void* stream);
/* Rehearsal instrument for one-GPU boxes (with one rank an all-reduce moves nothing): `workgroups` x 256 threads -- RCCL's launch
 * shape -- stream buf[0..count) `passes` times, read-modify-write with a factor of 1.0f (values unchanged), on `stream`: the
 * CUs and the HBM share a bucket's ring all-reduce would hold while the backward runs beside it.  No communicator involved;
 * buf 16-B aligned. */
int dlsg_comm_rehearsal(float* buf, int64_t count, int workgroups, int passes, void* stream);
/* *code <- ncclCommGetAsyncError of the communicator (0 = no error); does not synchronise. */
int dlsg_comm_async_error(dlsg_comm* c, int32_t* code);

#ifdef __cplusplus
}
#endif
#endif
