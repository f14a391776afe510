"""Forward / backward schedules of the CapGnnModel hot path over the kernel interface (`ops`, see hip.py).

The reference runs this path as ~8.7k eager ATen calls per forward.  Here the same
arithmetic is an explicit launch schedule over a handful of hand-written kernels:

  encoder  (models/layer.py:46-61,172-201; models/sublayer.py:63-82,189-198)
    * every projection is one fp32-MFMA GEMM; tanh is fused in the obj_embed epilogue
    * object->frame graph: one fused kernel (LayerNorm on the fly + scores + online softmax + aggregation)
    * LatentPSL / self-attention cores: batched GEMMs + a strided softmax kernel
    * BiLSTM: input gates for all 26 steps in one GEMM, then 26 x (grouped K-split GEMM + pointwise cell)
  decoder  (models/layer.py:394-462,569-602; models/sublayer.py:28-43)
    * step-invariant work hoisted out of the word loop: K' = (m W_K^T) W_Q, V' = (m W_V^T) W_O^T, the
      global-feature part of the query gates; so a step is 2 grouped GEMMs + 2 cell kernels + 1 attention
      kernel + 3 LayerNorm kernels, and the vocab projection of all 26 steps is ONE GEMM at the end
    * internal buffers are time-major (L, B, .) so weight gradients are single (L*B)-deep TN GEMMs

Backward is hand-scheduled (no autograd inside): activations the backward needs are kept in the `sv` dict.
Nothing here touches torch arithmetic: tensors are storage + strides for the kernels.
"""
import math

import torch

from .hip import GEMM_NT, GEMM_NN, GEMM_TN, F_ACCUM, F_TANH, F_BF16X3, F_FORCE128

V_SK = 7        # include/dlsg.h DLSG_GEMM_V_SK: the persistent stream-K kernel

# dropout sites (stateless masks are keyed by (seed, site, element index))
SITE_PSL_OBJ, SITE_PSL_MOT, SITE_LSTM, SITE_PE, SITE_SA, SITE_WORD, SITE_QUERY, SITE_ATT1, SITE_ATT2, SITE_LANG = range(1, 11)
# per-step sites add STEP_SITE*(t+1)
STEP_SITE = 64


def _empty(ref, *shape, dtype=torch.float32):
    return torch.empty(*shape, dtype=dtype, device=ref.device)


def _zeros(ref, *shape, dtype=torch.float32):
    return torch.zeros(*shape, dtype=dtype, device=ref.device)


def _ksplit_bounds(K, nsplit, align=32):
    """Split [0,K) into <= nsplit chunks whose boundaries are multiples of `align`: 32 = the GEMM K tile, 128 = the
    super-chunk of the skinny (M <= 64) kernel, whose partial tail chunks take the slower guarded-load path."""
    nsplit = max(1, min(nsplit, K // 512))       # a split shorter than ~512 costs more in launch/tail than it hides
    step = ((K + nsplit - 1) // nsplit + align - 1) // align * align
    out, k = [], 0
    while k < K:
        out.append((k, min(K, k + step)))
        k += step
    return out


SKINNY_M = 128          # batch rows the weight-streaming (skinny) GEMM kernels take: 64- or 128-row tiles (csrc/gemm.hip)


def _bwd_bounds(K, M, nsplit):
    """K-split of an input-gradient product dy (M, K) @ W (K, N).  For the skinny kernels (M <= 128) 1024-deep chunks measured
    best on the batch-64 step (4 x 1024 for K = 4096: +0.6 % step throughput over 6 x 768 / 8 x 512; 5 x 896 is 2.7 % worse)."""
    if M <= SKINNY_M and K >= 2048:
        return [(k, min(K, k + 1024)) for k in range(0, K, 1024)]
    return _ksplit_bounds(K, nsplit, 128 if M <= SKINNY_M else 32)


def _nsplit_for(M, N, nseg, target=384, cap=16):
    # (a skinny launch has ONE row tile whatever the batch: its workgroups are column tiles x groups)
    tiles = (1 if M <= SKINNY_M else (M + 63) // 64) * ((N + 63) // 64)
    n = max(1, target // max(1, tiles * nseg))
    return max(1, min(n, cap // max(1, nseg)))


def seg_gemm_nt(ops, segs, M, N, ref):
    """sum_i  x_i @ W_i^T  as ONE grouped launch writing K-split slabs.  segs: list of (x (M,K_i), W (N,K_i)).
    Returns slabs (S, M, N); the consumer kernel (lstm_pw / slab_reduce) sums them."""
    # M <= 128 (skinny kernels): one group per input segment -- whole 1024-deep segments measured better than 512-deep halves
    ns = 1 if M <= SKINNY_M else _nsplit_for(M, N, len(segs))
    groups = []
    for x, W in segs:
        for k0, k1 in _ksplit_bounds(x.shape[1], ns, 128 if M <= SKINNY_M else 32):
            groups.append((x[:, k0:k1], W[:, k0:k1]))
    slabs = _empty(ref, len(groups), M, N)
    ops.gemm(GEMM_NT, [(a, b, slabs[i]) for i, (a, b) in enumerate(groups)])
    return slabs


def gemm_nn_split(ops, dy, W, out, ref, accum=False):
    """out (M, Kin) (+)= dy (M, Nout) @ W (Nout, Kin) with the contraction split over slabs (small M)."""
    M, Nn = dy.shape
    Kin = W.shape[1]
    ns = _nsplit_for(M, Kin, 1)
    bounds = _ksplit_bounds(Nn, ns, 128 if M <= SKINNY_M else 32)
    if len(bounds) == 1:
        ops.gemm(GEMM_NN, [(dy, W, out)], flags=F_ACCUM if accum else 0)
        return
    slabs = _empty(ref, len(bounds), M, Kin)
    ops.gemm(GEMM_NN, [(dy[:, k0:k1], W[k0:k1, :], slabs[i]) for i, (k0, k1) in enumerate(bounds)])
    ops.slab_reduce(slabs, out, flags=F_ACCUM if accum else 0)


DEEP_TN_CHUNKS = 8


def _accum_flag(ops, gout):
    """F_ACCUM unless this is the first product written into `gout` since the gradient arena was zero-filled (the backward sets
    ops.grad_written = set() right after the fill): a first writer stores instead of adding, which spares the kernel's epilogue a
    read of zeros in front of every store.  Without that bookkeeping (ops.grad_written is None) everything accumulates."""
    seen = getattr(ops, 'grad_written', None)
    if seen is None:
        return F_ACCUM
    key = (gout.data_ptr(), tuple(gout.shape), tuple(gout.stride()))
    if key in seen:
        return F_ACCUM
    seen.add(key)
    return 0


def _accum_always(ops, gout):
    """F_ACCUM for a writer that always adds (it is not known to be the block's first), and the block is marked as written: a
    tracked writer that comes later must not take itself for the first and store over this contribution"""
    seen = getattr(ops, 'grad_written', None)
    if seen is not None:
        seen.add((gout.data_ptr(), tuple(gout.shape), tuple(gout.stride())))
    return F_ACCUM


def gemm_tn_deep(ops, items, ref):
    """gout_i (Nout, Kin) += dy_i^T x_i for very deep contractions (rows >= 8192: the 26624-row obj_embed weight gradients):
    the output has too few tiles to fill the chip, so the rows are split over groups writing slabs (measured 98 vs 86
    TFLOP/s in fp32; in the step, also 0.5 ms faster than un-split on the split-bf16 path) and the slabs are folded into
    the gradient.  Several products (the two streams' obj_embed) share ONE launch: 8 row groups x 128 tiles is 1024
    workgroups on 768 slots -- a second round one third full; two products together are 2048 = 2 2/3 rounds, not 4."""
    deep = [it for it in items if it[0].shape[0] >= 8192]
    for dy, x, gout in items:
        if dy.shape[0] < 8192:
            ops.gemm(GEMM_TN, [(dy, x, gout)], flags=_accum_flag(ops, gout))
    if deep and getattr(ops, 'stream_k', False) and all(ops.gemm(GEMM_TN, [it], flags=F_ACCUM, plan_only=True) == V_SK for it in deep):
        # the persistent stream-K kernel cuts the contraction between the workgroups itself (csrc/gemm_sk.hip): one launch for
        # all the products, no slabs, no fold
        parts = {0: [], F_ACCUM: []}
        for it in deep:
            parts[_accum_flag(ops, it[2])].append(it)
        for fl, part in parts.items():
            for i0 in range(0, len(part), 16):
                ops.gemm(GEMM_TN, part[i0:i0 + 16], flags=fl)
        return
    by_shape = {}
    for it in deep:
        by_shape.setdefault((it[0].shape, it[1].shape), []).append(it)
    ks = DEEP_TN_CHUNKS
    per_launch = max(1, min(2, 16 // ks))
    for grp in by_shape.values():
        for i0 in range(0, len(grp), per_launch):
            part = grp[i0:i0 + per_launch]
            rows = part[0][0].shape[0]
            # ceil, not floor: rows = 8450 (5 objects x 26 frames x 65 clips) with floor gave 9 chunks -> 18 groups > MAXG
            step = ((rows + ks - 1) // ks + 31) // 32 * 32
            bounds = [(k, min(rows, k + step)) for k in range(0, rows, step)]
            assert len(bounds) * len(part) <= 16
            slabs = [_empty(ref, len(bounds), gout.shape[0], gout.shape[1]) for _, _, gout in part]
            ops.gemm(GEMM_TN, [(dy[k0:k1], x[k0:k1], sl[i]) for (dy, x, _), sl in zip(part, slabs)
                               for i, (k0, k1) in enumerate(bounds)])
            for (_, _, gout), sl in zip(part, slabs):
                ops.slab_reduce(sl, gout, flags=_accum_always(ops, gout))


def _accum_flag_peek(ops, gout):
    seen = getattr(ops, 'grad_written', None)
    if seen is None or (gout.data_ptr(), tuple(gout.shape), tuple(gout.stride())) in seen:
        return F_ACCUM
    return 0


def gemm_nn_multi(ops, dy, Ws, out, ref):
    """out (M, sum Kin_i) = [dy @ W_0 | dy @ W_1 | ...] for several weight blocks sharing the same dy (e.g. W_ih and W_hh
    of one cell): ONE grouped launch (groups have their own output width) writing column blocks of one slab set, ONE
    slab_reduce.  dy (M, Nout), W_i (Nout, Kin_i)."""
    M, Nn = dy.shape
    widths = [W.shape[1] for W in Ws]
    tot = sum(widths)
    assert out.shape[1] == tot
    ns = _nsplit_for(M, tot, 1)
    ns = max(1, min(ns, 16 // len(Ws)))
    bounds = _bwd_bounds(Nn, M, ns)
    slabs = _empty(ref, len(bounds), M, tot)
    groups, c0 = [], 0
    for W, wd in zip(Ws, widths):
        for i, (k0, k1) in enumerate(bounds):
            groups.append((dy[:, k0:k1], W[k0:k1, :], slabs[i][:, c0:c0 + wd]))
        c0 += wd
    ops.gemm(GEMM_NN, groups)
    ops.slab_reduce(slabs, out)


def gemm_nn_multi_slabs(ops, dy, Ws, width, ref):
    """gemm_nn_multi without the reduction: returns the slab stack (S, M, width) whose sum over S is
    [dy @ W_0 | dy @ W_1 | ...] in the leading columns (the consumer kernel sums the slabs while it reads them)."""
    M, Nn = dy.shape
    widths = [W.shape[1] for W in Ws]
    tot = sum(widths)
    assert tot <= width
    ns = _nsplit_for(M, tot, 1)
    ns = max(1, min(ns, 16 // len(Ws)))
    bounds = _bwd_bounds(Nn, M, ns)
    if M <= SKINNY_M and len(set(widths)) > 1:
        # recurrent products: a launch whose groups are all equally wide runs on 64-column workgroups, one per CU (csrc/gemm.hip,
        # launch_skinny_mi) -- the wide weight is cut into column blocks as wide as the narrow one when that stays within 16 groups
        # (the language cell's [W_ih 3072 | W_hh 1024] over four K chunks: 16 groups of 1024 columns)
        wmin = min(widths)
        if all(wd % wmin == 0 for wd in widths) and (tot // wmin) * len(bounds) <= 16:
            Ws = [W[:, c:c + wmin] for W, wd in zip(Ws, widths) for c in range(0, wd, wmin)]
            widths = [wmin] * len(Ws)
    slabs = _empty(ref, len(bounds), M, width)
    groups, c0 = [], 0
    for W, wd in zip(Ws, widths):
        for i, (k0, k1) in enumerate(bounds):
            groups.append((dy[:, k0:k1], W[k0:k1, :], slabs[i][:, c0:c0 + wd]))
        c0 += wd
    ops.gemm(GEMM_NN, groups)
    return slabs


def tn_grouped(ops, items, defer=None):
    """Weight gradients gout_i += dy_i^T x_i for several (dy, x, gout) triples: one grouped launch per distinct output
    height (dy width) instead of one launch each.  A 4096 x 1024 gradient block is 256 tiles of 128 x 128 -- one per CU,
    nothing to hide latency behind; seven of them in one launch fill every CU three deep and run on the large tile.
    defer: a list collecting the triples instead (one process, no bucket cuts: the backward launches ALL its mid-size weight
    gradients at its end, grouped by height across modules -- the decoder's and the BiLSTM's 4096-row blocks are 2.3 + 1.3
    rounds of workgroups apart and 3.7 together; worth 0.04 ms of the 15.05-ms step: a partly filled round's workgroups run
    faster, so little was lost to begin with)."""
    if defer is not None:
        defer.extend(items)
        return
    # (one launch = one height AND one depth: the stream-K kernel cuts its remainder tiles between the workgroups only when they
    #  are equally deep -- 1 664-deep and 512-deep blocks in one launch ran 0.52 ms instead of 0.37 + 0.09)
    by_m = {}
    for dy, x, gout in items:
        by_m.setdefault((dy.shape[1], dy.shape[0]), []).append((dy, x, gout))
    for grp in by_m.values():
        # two triples writing the same gradient block must not share a launch (accumulating groups would race)
        waves, seen = [[]], [set()]
        for it in grp:
            key = (it[2].data_ptr(), tuple(it[2].shape))
            for w_, sn in zip(waves, seen):
                if key not in sn and len(w_) < 16:
                    w_.append(it); sn.add(key)
                    break
            else:
                waves.append([it]); seen.append({key})
        for w_ in waves:
            # first writers store, the rest accumulate: two launches at most
            first = [it for it in w_ if _accum_flag_peek(ops, it[2]) == 0]
            rest = [it for it in w_ if _accum_flag_peek(ops, it[2]) != 0]
            for it in first:
                _accum_flag(ops, it[2])
            if first:
                ops.gemm(GEMM_TN, first, flags=0)
            if rest:
                ops.gemm(GEMM_TN, rest, flags=F_ACCUM)


def gemm_bucketed(ops, mode, groups, flags=0):
    """groups (A, B, C) of one mode launched together whenever their output heights agree (the two attention streams,
    the three projections of the self-attention, ...): one launch per distinct C row count."""
    by_m = {}
    for g in groups:
        by_m.setdefault(g[2].shape[-2], []).append(g)
    for grp in by_m.values():
        for i in range(0, len(grp), 16):
            ops.gemm(mode, grp[i:i + 16], flags=flags)


def lin(ops, x, W, out, bias=None, tanh=False, accum=False, skip_if=None):
    ops.gemm(GEMM_NT, [(x, W, out)], flags=(F_TANH if tanh else 0) | (F_ACCUM if accum else 0), bias=bias, skip_if=skip_if)


class Grads(object):
    """name -> gradient view (all views of one flat fp32 arena, zero-filled at the start of a backward)."""

    def __init__(self, named_params, arena, offsets):
        self.arena, self.offsets = arena, offsets
        self.views = {}
        for name, p in named_params:
            o = offsets[name]
            self.views[name] = arena[o:o + p.numel()].view(p.shape)

    def __getitem__(self, name):
        return self.views[name]


def ln_grads(ops, part, G, prefix, n):
    """fold rowln_bwd partials (nblk, 2, n) into the LayerNorm weight/bias gradients."""
    p2 = part.view(-1, 2 * n)
    ops.colsum2(p2, G[prefix + '.weight'], G[prefix + '.bias'], split=n, accum=True)


# ================================================================================================ TUN encoder
def region_projections(ops, mods, regions):
    """y_i = tanh(obj_embed_i(regions)) for every stream in ONE grouped launch (layer.py:184 runs once per stream on the
    same regions): the two 1664-tile grids of the batch-64 step pack into the chip's 768 workgroup slots in 5 rounds
    instead of 3 + 3, and the second stream finds the region panel in L2."""
    B, T, O, R = regions.shape
    rows = B * T * O
    r2 = regions.view(rows, R)
    ys = [_empty(regions, rows, m.obj_embed.weight.shape[0]) for m in mods]
    # (wave quantisation is not what this launch loses: main rows in whole 768-slot rounds + the tail split along K measured
    #  1.78 ms against 1.75 ms for the one launch, round 3)
    ops.gemm(GEMM_NT, [(r2, m.obj_embed.weight, y, m.obj_embed.bias) for m, y in zip(mods, ys)], flags=F_TANH)
    return ys


def tun_fwd(ops, m, pfx, visual, regions, sv, training, seed, psl_site, fused_o2v=True, nsplit=None, y=None):
    """EncoderVisualGraphTUN.forward (models/layer.py:172-201).  visual: (B*T, Hin) view."""
    tun_frames(ops, m, pfx, visual, regions, sv)
    tun_graph(ops, [(m, pfx, y)], regions, sv, fused_o2v, nsplit)
    return tun_latent(ops, m, pfx, regions, sv, training, seed, psl_site)


def tun_frames(ops, m, pfx, visual, regions, sv):
    """frame nodes v = LN(tanh(visual_embed(x))) (layer.py:179-180; the Linear is skipped when use_embed is False)"""
    tun_frames_multi(ops, [(m, pfx, visual)], regions, sv)


def tun_frames_multi(ops, items, regions, sv, pre=None):
    """tun_frames of several streams; items: [(module, prefix, visual)].  Streams whose frame nodes have one width share ONE
    visual_embed launch (those that have the Linear) and ONE visual_norm launch.  pre: {prefix: v_pre} for streams whose
    visual_embed output exists already."""
    B, T, O, R = regions.shape
    ref = regions
    Hs = [m.visual_norm[1].weight.numel() for m, _, _ in items]
    pre = pre or {}
    if len(items) > 1 and len(set(Hs)) > 1:
        for it in items:
            tun_frames_multi(ops, [it], regions, sv, pre)
        return
    H = Hs[0]
    groups, calls = [], []
    for m, pfx, visual in items:
        s = sv[pfx] = {}
        if m.use_embed and pfx in pre:
            v_pre = pre[pfx]                   # visual_embed already ran (as a group of another launch)
        elif m.use_embed:
            v_pre = _empty(ref, B * T, H)
            groups.append((visual, m.visual_embed.weight, v_pre, m.visual_embed.bias))
        else:
            v_pre = visual
        v = _empty(ref, B * T, H); st_v = _empty(ref, B * T, 2)
        calls.append(dict(x=v_pre, gamma=m.visual_norm[1].weight, beta=m.visual_norm[1].bias, y=v, stats=st_v, pre_tanh=1))
        s.update(visual=visual, v_pre=v_pre, v=v, st_v=st_v)
    if groups:
        ops.gemm(GEMM_NT, groups)
    ops.rowln_fwd_multi(calls)


O2V_MAX_NSPLIT = 64      # csrc/attention.hip: dlsg_o2v_fwd_multi / dlsg_o2v_bwd refuse more chunks per clip


def o2v_nsplit(B_eff, NO):
    """object chunks per clip of the fused graph kernel: one 156-KB-LDS workgroup per CU walks 16-object tiles; split a
    clip's objects only as far as needed to put ~256 workgroups on the chip (every extra split costs a (T x H) partial
    written and re-read).  B_eff = clips x streams of the launch."""
    tiles = (NO + 15) // 16
    return max(1, min(tiles, 256 // max(B_eff, 1), O2V_MAX_NSPLIT))


def tun_graph(ops, items, regions, sv, fused_o2v=True, nsplit=None):
    """object -> frame graph (layer.py:184-192) of one or several streams; items: [(module, prefix, y or None)].  Streams of
    one shape share ONE fused launch (CapGnnEncoder: 2 x 64 clips fill the chip with two object chunks per clip instead
    of four).  Leaves z = agg + v (pre obj_visual_norm) and the saved statistics in sv[prefix]."""
    B, T, O, R = regions.shape
    if O < 5:
        return
    NO = T * O
    ref = regions
    scale = 1.0 / math.sqrt(R)
    fused, plain = [], []
    for m, pfx, y in items:
        s = sv[pfx]
        H = m.visual_norm[1].weight.numel()
        if y is None:
            y = _empty(ref, B * NO, H)
            lin(ops, regions.view(B * NO, R), m.obj_embed.weight, y, m.obj_embed.bias, tanh=True)
        z = _empty(ref, B * T, H); ostats = _empty(ref, B * NO, 2); S = _empty(ref, B, NO, T)
        s.update(y=y, z=z, ostats=ostats, S=S, scale=scale)
        (fused if fused_o2v and ops.o2v_supported(T, H) else plain).append((m, pfx, H))
    by_h = {}
    for it in fused:
        by_h.setdefault(it[2], []).append(it)
    for H, grp in by_h.items():
        for i0 in range(0, len(grp), 2):
            part = grp[i0:i0 + 2]
            ns = o2v_nsplit(B * len(part), NO) if nsplit is None else nsplit
            args = []
            for m, pfx, _ in part:
                s = sv[pfx]
                s['ml'] = _empty(ref, B * T, 2)
                args.append(dict(y=s['y'].view(B, NO, H), v=s['v'].view(B, T, H), g_obj=m.obj_norm[1].weight,
                                 b_obj=m.obj_norm[1].bias, z=s['z'], ml=s['ml'], ostats=s['ostats'], S=s['S']))
            ops.o2v_fwd_multi(args, scale, ns)
    for m, pfx, H in plain:
        s = sv[pfx]
        y, v, z, S = s['y'], s['v'], s['z'], s['S']
        o = _empty(ref, B * NO, H)
        ops.rowln_fwd(y, m.obj_norm[1].weight, m.obj_norm[1].bias, o, s['ostats'])
        ops.gemm(GEMM_NT, [(o.view(B, NO, H), v.view(B, T, H), S)], alpha=scale)
        Pm = _empty(ref, B, NO, T)
        ops.softmax_fwd(S, Pm, B, NO, T)
        ops.copy2d(v, z)
        ops.gemm(GEMM_TN, [(Pm, o.view(B, NO, H), z.view(B, T, H))], flags=F_ACCUM)


def _same_shapes(mods):
    m0 = mods[0]
    return all(m.visual_norm[1].weight.numel() == m0.visual_norm[1].weight.numel() and m.v2l_layer.theta.shape == m0.v2l_layer.theta.shape
               and m.baseline == m0.baseline for m in mods)


def tun_latent(ops, m, pfx, regions, sv, training, seed, psl_site):
    """ov = LN(tanh(agg + v)) (layer.py:192-193) and LatentPSL (layer.py:199, sublayer.py:189-198); baseline streams return ov."""
    return tun_latent_multi(ops, [(m, pfx, psl_site)], regions, sv, training, seed)[0]


def tun_latent_multi(ops, items, regions, sv, training, seed):
    """tun_latent of several streams; items: [(module, prefix, psl_site)].  Streams of one shape share ONE obj_visual_norm launch and
    ONE LatentPSL launch (CapGnnEncoder: one workgroup per clip -- 64 clips of one stream leave three quarters of the chip idle)."""
    if len(items) > 1 and not _same_shapes([it[0] for it in items]):
        return [tun_latent_multi(ops, [it], regions, sv, training, seed)[0] for it in items]
    B, T, O, R = regions.shape
    ref = regions
    m0 = items[0][0]
    H = m0.visual_norm[1].weight.numel()
    P = m0.v2l_layer.theta.shape[0]
    ovs = []
    if O >= 5:
        calls = []
        for m, pfx, _ in items:
            s = sv[pfx]
            ov = _empty(ref, B * T, H); st_ov = _empty(ref, B * T, 2)
            calls.append(dict(x=s['z'], gamma=m.obj_visual_norm[1].weight, beta=m.obj_visual_norm[1].bias, y=ov, stats=st_ov, pre_tanh=1))
            s.update(ov=ov, st_ov=st_ov)
            ovs.append(ov)
        ops.rowln_fwd_multi(calls)
    else:
        for m, pfx, _ in items:
            sv[pfx].update(ov=sv[pfx]['v'])
            ovs.append(sv[pfx]['v'])
    if m0.baseline:
        return ovs
    # LatentPSL (models/sublayer.py:189-198)
    pd = 0.3 if training else 0.0
    outs, calls = [], []
    fused = ops.latent_psl_supported(T, P, H)
    for (m, pfx, psl_site), ov in zip(items, ovs):
        s = sv[pfx]
        theta = m.v2l_layer.theta
        adj = _empty(ref, B, T, P)
        u = _empty(ref, B * P, H)
        psl = _empty(ref, B * P, H); st_p = _empty(ref, B * P, 2)
        ln = m.v2l_layer.out_norm[1]
        if fused:
            calls.append(dict(ov=ov.view(B, T, H), theta=theta, gamma=ln.weight, beta=ln.bias, adj=adj, u=u, out=psl, stats=st_p, p=pd,
                              site=psl_site, seed=seed))
        else:
            lg = _empty(ref, B, T, P)
            ops.gemm(GEMM_NT, [(ov.view(B, T, H), theta.unsqueeze(0).expand(B, P, H), lg)])
            ops.softmax_fwd(lg, adj, B, T, P)
            ops.gemm(GEMM_TN, [(adj, ov.view(B, T, H), u.view(B, P, H))])
            ops.rowln_fwd(u, ln.weight, ln.bias, psl, st_p, pre_tanh=1, p1=pd, site1=psl_site, seed=seed)
        s.update(adj=adj, u=u, st_p=st_p, pd=pd, psl_site=psl_site)
        outs.append(psl.view(B, P, H))
    if calls:
        ops.latent_psl_fwd_multi(calls)
    return outs


def tun_bwd(ops, m, pfx, regions, sv, G, dpsl, training, seed, defer_dw=None):
    """Backward of tun_fwd.  dpsl (B,P,H).  Returns d(visual input) when the stream has no embed, else None."""
    tun_bwd_head(ops, m, pfx, regions, sv, G, dpsl, training, seed)
    tun_graph_bwd(ops, [(m, pfx)], regions, sv, G)
    return tun_bwd_tail(ops, m, pfx, regions, sv, G, defer_dw)


def tun_bwd_head(ops, m, pfx, regions, sv, G, dpsl, training, seed):
    """LatentPSL and obj_visual_norm backward: leaves dz = d(agg + v) in sv[pfx] (or d(v) directly when the graph is skipped)."""
    tun_bwd_head_multi(ops, [(m, pfx, dpsl)], regions, sv, G, training, seed)


def tun_bwd_head_multi(ops, items, regions, sv, G, training, seed):
    """tun_bwd_head of several streams; items: [(module, prefix, dpsl)].  Streams of one shape share ONE launch of the LatentPSL
    backward and ONE of the obj_visual_norm backward."""
    if len(items) > 1 and not _same_shapes([it[0] for it in items]):
        for it in items:
            tun_bwd_head_multi(ops, [it], regions, sv, G, training, seed)
        return
    B, T, O, R = regions.shape
    ref = regions
    m0 = items[0][0]
    H = m0.visual_norm[1].weight.numel()
    P = m0.v2l_layer.theta.shape[0]
    dovs = []
    if m0.baseline:
        # baseline streams return the frame nodes ov themselves (layer.py:197-198): dpsl is d(ov), (B,T,H)
        dovs = [dpsl.reshape(B * T, H) for _, _, dpsl in items]
    elif ops.latent_psl_bwd_supported(T, P, H):
        calls, after = [], []
        for m, pfx, dpsl in items:
            s = sv[pfx]
            ln = m.v2l_layer.out_norm[1]
            dov = _empty(ref, B * T, H)
            part = _empty(ref, B, 2, H)
            dth = _empty(ref, B, P * H)
            calls.append(dict(dout=dpsl.reshape(B * P, H), u=s['u'], stats=s['st_p'], gamma=ln.weight, adj=s['adj'], ov=s['ov'].view(B, T, H),
                              theta=m.v2l_layer.theta, dov=dov, dtheta_part=dth.view(B, P, H), part=part, p=s['pd'], site=s['psl_site'],
                              seed=seed))
            after.append((pfx, part, dth))
            dovs.append(dov)
        ops.latent_psl_bwd_multi(calls)
        for pfx, part, dth in after:
            ln_grads(ops, part, G, pfx + '.v2l_layer.out_norm.1', H)
            ops.colsum(dth, G[pfx + '.v2l_layer.theta'].view(P * H), accum=True)
    else:
        for m, pfx, dpsl in items:
            s = sv[pfx]
            ln = m.v2l_layer.out_norm[1]
            ov, adj, theta = s['ov'], s['adj'], m.v2l_layer.theta
            dov = _empty(ref, B * T, H)
            nb = ops.rowln_bwd_nblk(B * P)
            part = _empty(ref, nb, 2, H)
            du = _empty(ref, B * P, H)
            ops.rowln_bwd(dpsl.reshape(B * P, H), s['u'], ln.weight, ln.bias, du, stats=s['st_p'], pre_tanh=1, p1=s['pd'],
                          site1=s['psl_site'], seed=seed, dgb_part=part)
            ln_grads(ops, part, G, pfx + '.v2l_layer.out_norm.1', H)
            du3 = du.view(B, P, H)
            dadj = _empty(ref, B, T, P)
            ops.gemm(GEMM_NT, [(ov.view(B, T, H), du3, dadj)])
            ops.gemm(GEMM_NN, [(adj, du3, dov.view(B, T, H))])
            dlg = _empty(ref, B, T, P)
            ops.softmax_bwd(adj, dadj, dlg, B, T, P)
            ops.gemm(GEMM_NN, [(dlg, theta.unsqueeze(0).expand(B, P, H), dov.view(B, T, H))], flags=F_ACCUM)
            ops.gemm(GEMM_TN, [(dlg.view(B * T, P), ov, G[pfx + '.v2l_layer.theta'])], flags=_accum_always(ops, G[pfx + '.v2l_layer.theta']))
            dovs.append(dov)
    if O >= 5:
        nb = ops.rowln_bwd_nblk(B * T)
        calls, after = [], []
        for (m, pfx, _), dov in zip(items, dovs):
            s = sv[pfx]
            lnv = m.obj_visual_norm[1]
            part = _empty(ref, nb, 2, H)
            dz = _empty(ref, B * T, H)
            calls.append(dict(dy=dov, x=s['z'], gamma=lnv.weight, beta=lnv.bias, dx=dz, stats=s['st_ov'], pre_tanh=1, dgb_part=part))
            after.append((pfx, part))
            s['dz'] = dz
        ops.rowln_bwd_multi(calls)
        for pfx, part in after:
            ln_grads(ops, part, G, pfx + '.obj_visual_norm.1', H)
    else:
        for (m, pfx, _), dov in zip(items, dovs):
            sv[pfx]['dv'] = dov


def tun_graph_bwd(ops, items, regions, sv, G):
    """Backward of the object -> frame graph (layer.py:184-192) of one or several streams; items: [(module, prefix)].  Streams
    whose forward ran the fused kernel share ONE launch per pass (csrc/o2v16_bwd.hip).  Leaves dy (gradient w.r.t. the
    obj_embed pre-activation) and dv in sv[prefix]; folds obj_norm's dgamma / dbeta."""
    B, T, O, R = regions.shape
    if O < 5:
        return
    NO = T * O
    ref = regions
    by_h = {}
    for m, pfx in items:
        if 'ml' in sv[pfx]:
            by_h.setdefault(m.visual_norm[1].weight.numel(), []).append((m, pfx))
    for H, grp in by_h.items():
        for i0 in range(0, len(grp), 2):
            part_items = grp[i0:i0 + 2]
            ns = o2v_nsplit(B * len(part_items), NO)
            args = []
            for m, pfx in part_items:
                s = sv[pfx]
                s['dy'] = _empty(ref, B * NO, H)
                s['dv'] = _empty(ref, B * T, H)
                # (the second pass also leaves the column sums of dy per (clip, chunk): obj_embed's bias gradient from B * ns
                #  rows instead of another pass over the B * NO rows of dy -- 218 MB per stream at batch 64)
                s['dysum'] = _empty(ref, B * ns, H)
                args.append(dict(y=s['y'].view(B, NO, H), ostats=s['ostats'], g_obj=m.obj_norm[1].weight, b_obj=m.obj_norm[1].bias,
                                 v=s['v'].view(B, T, H), z=s['z'].view(B, T, H), dz=s['dz'].view(B, T, H), S=s['S'], ml=s['ml'],
                                 dy=s['dy'].view(B, NO, H), dv=s['dv'].view(B, T, H), dysum=s['dysum']))
            parts = ops.o2v_bwd_multi(args, sv[part_items[0][1]]['scale'], ns)
            for (m, pfx), part in zip(part_items, parts):
                ln_grads(ops, part, G, pfx + '.obj_norm.1', H)
    for m, pfx in items:
        s = sv[pfx]
        if 'ml' in s:
            continue
        H = m.visual_norm[1].weight.numel()
        g_o, b_o = m.obj_norm[1].weight, m.obj_norm[1].bias
        y, v, S, scale, dz = s['y'], s['v'], s['S'], s['scale'], s['dz']
        o = _empty(ref, B * NO, H)
        ops.rowln_fwd(y, g_o, b_o, o, None)
        o3, dz3, v3 = o.view(B, NO, H), dz.view(B, T, H), v.view(B, T, H)
        Pm = _empty(ref, B, NO, T)
        ops.softmax_fwd(S, Pm, B, NO, T)
        dP = _empty(ref, B, NO, T)
        ops.gemm(GEMM_NT, [(o3, dz3, dP)])
        dS = _empty(ref, B, NO, T)
        ops.softmax_bwd(Pm, dP, dS, B, NO, T)
        do = _empty(ref, B * NO, H)
        ops.gemm(GEMM_NN, [(Pm, dz3, do.view(B, NO, H))])
        ops.gemm(GEMM_NN, [(dS, v3, do.view(B, NO, H))], alpha=scale, flags=F_ACCUM)
        dv = _empty(ref, B * T, H)
        ops.copy2d(dz, dv)
        ops.gemm(GEMM_TN, [(dS, o3, dv.view(B, T, H))], alpha=scale, flags=F_ACCUM)
        nb = ops.rowln_bwd_nblk(B * NO)
        part = _empty(ref, nb, 2, H)
        dy = o   # reuse the scratch: rowln_bwd reads y/stats, not o
        ops.rowln_bwd(do, y, g_o, b_o, dy, stats=s['ostats'], pre_tanh=2, dgb_part=part)
        ln_grads(ops, part, G, pfx + '.obj_norm.1', H)
        s['dy'], s['dv'] = dy, dv


def tun_bwd_tail(ops, m, pfx, regions, sv, G, defer_dw=None):
    """obj_embed and frame-node gradients.  Returns d(visual input) when the stream has no embed, else None."""
    B, T, O, R = regions.shape
    s = sv[pfx]
    H = m.visual_norm[1].weight.numel()
    ref = regions
    name = pfx
    if O >= 5:
        NO = T * O
        dy = s['dy']
        if defer_dw is not None:
            defer_dw.append((dy, regions.view(B * NO, R), G[name + '.obj_embed.weight']))
        else:
            gemm_tn_deep(ops, [(dy, regions.view(B * NO, R), G[name + '.obj_embed.weight'])], ref)
        ops.colsum(s['dysum'] if s.get('dysum') is not None else dy, G[name + '.obj_embed.bias'], accum=True)
    dv = s['dv']
    lnv = m.visual_norm[1]
    nb = ops.rowln_bwd_nblk(B * T)
    part = _empty(ref, nb, 2, H)
    dv_pre = _empty(ref, B * T, H)
    ops.rowln_bwd(dv, s['v_pre'], lnv.weight, lnv.bias, dv_pre, stats=s['st_v'], pre_tanh=1, dgb_part=part)
    ln_grads(ops, part, G, name + '.visual_norm.1', H)
    if m.use_embed:
        tn_grouped(ops, [(dv_pre, s['visual'], G[name + '.visual_embed.weight'])], sv.get('tn_defer'))
        ops.colsum(dv_pre, G[name + '.visual_embed.bias'], accum=True)
        return None
    return dv_pre


# ================================================================================================ EncoderVisual
BILSTM_ROWS = 64        # batch rows of one persistent BiLSTM launch (csrc/bilstm.hip BL_ROWS)


def _bilstm_steps_fwd(ops, xg, Whh, bih, bhh, out, hprev, cst, gates, B, T, H, ref):
    """the BiLSTM recurrence step by step: per step one grouped K-split skinny GEMM (both directions) + one pointwise launch"""
    ns = _nsplit_for(B, 4 * H, 2)
    bounds = _ksplit_bounds(H, ns, 128 if B <= SKINNY_M else 32)
    for step in range(T):
        tt = [step, T - 1 - step]
        tp = [step - 1, T - step]                                # previous time index per direction
        slabs = None
        if step > 0:
            slabs = _empty(ref, 2 * len(bounds), B, 4 * H)
            groups = []
            for d in range(2):
                hp = hprev[d][:, tt[d]]
                for i, (k0, k1) in enumerate(bounds):
                    groups.append((hp[:, k0:k1], Whh[d][:, k0:k1], slabs[d * len(bounds) + i]))
            ops.gemm(GEMM_NT, groups)
        calls = []
        for d in range(2):
            t = tt[d]
            nxt = t + 1 if d == 0 else t - 1
            h2 = hprev[d][:, nxt] if 0 <= nxt < T else None
            calls.append(dict(slabs=slabs[d * len(bounds):(d + 1) * len(bounds)] if slabs is not None else None,
                              c=cst[d][:, t], B=B, H=H, addend=xg[d].view(B, T, 4 * H)[:, t], b_ih=bih[d], b_hh=bhh[d],
                              c_prev=cst[d][:, tp[d]] if step > 0 else None, h=out[:, t, d * H:(d + 1) * H], h2=h2,
                              gates=gates[d][:, t]))
        ops.lstm_pw_fwd_multi(calls)            # both directions: one launch


def encvis_fwd(ops, m, pfx, frames2d, B, T, sv, training, seed, extra=None):
    """EncoderVisual.forward (models/layer.py:46-61).  frames2d (B*T, A+M).  Returns (B*T, H).
    extra: independent NT products (x, W, out, bias) on the same B*T rows that ride in the linear_embed launch (the object stream's
    visual_embed: 82 us as a launch of its own on the small tiles, ~58 us as one more group of the stream-K launch)."""
    H = m.hidden_size
    ref = frames2d
    s = sv[pfx] = {}
    e = _empty(ref, B * T, H)
    Kf = frames2d.shape[1]
    extra = list(extra or [])
    main = [(frames2d, m.linear_embed.weight, e, m.linear_embed.bias)]
    # (only products as deep as linear_embed's may share its launch: the stream-K kernel deals out equally deep tiles)
    ride = [g for g in extra if g[0].shape[1] == Kf]
    extra = [g for g in extra if g[0].shape[1] != Kf]
    if getattr(ops, 'stream_k', False) and ops.gemm(GEMM_NT, main + ride, plan_only=True) == V_SK:
        # one persistent stream-K launch cuts the 6 144-deep contraction between the workgroups itself (178 us against 227)
        ops.gemm(GEMM_NT, main + ride)
    elif B * T <= 2048 and Kf >= 4096 and Kf % 96 == 0:
        extra = ride + extra
        # a 1 664 x 1 024 output is 416 tiles for a 6 144-deep contraction: three K thirds as groups writing slabs + one fold
        # (210 us against 228, tools/_exp: K-split probe of the mid-size products)
        k3 = Kf // 3
        sl = _empty(ref, 3, B * T, H)
        ops.gemm(GEMM_NT, [(frames2d[:, i * k3:(i + 1) * k3], m.linear_embed.weight[:, i * k3:(i + 1) * k3], sl[i]) for i in range(3)])
        ops.slab_reduce(sl, e, bias=m.linear_embed.bias)
    else:
        extra = ride + extra
        lin(ops, frames2d, m.linear_embed.weight, e, m.linear_embed.bias)
    if extra and m.baseline:
        ops.gemm(GEMM_NT, extra)
        extra = []
    lstm = m.lstm
    Wih = [lstm.weight_ih_l0, lstm.weight_ih_l0_reverse]
    Whh = [lstm.weight_hh_l0, lstm.weight_hh_l0_reverse]
    bih = [lstm.bias_ih_l0, lstm.bias_ih_l0_reverse]
    bhh = [lstm.bias_hh_l0, lstm.bias_hh_l0_reverse]
    xg = [_empty(ref, B * T, 4 * H), _empty(ref, B * T, 4 * H)]
    ops.gemm(GEMM_NT, [(e, Wih[0], xg[0]), (e, Wih[1], xg[1])])
    out = _empty(ref, B, T, 2 * H)
    hz = _zeros(ref, 2, B, T, H)
    hprev = [hz[0], hz[1]]                                     # h of the previous step of each direction
    cst = [_empty(ref, B, T, H), _empty(ref, B, T, H)]
    gates = [_empty(ref, B, T, 4 * H), _empty(ref, B, T, 4 * H)]
    if getattr(ops, 'bilstm_supported', None) is not None and ops.bilstm_supported(min(B, BILSTM_ROWS), T, H):
        # the whole recurrence of both directions as ONE persistent launch: W_hh slices resident in LDS, h_t exchanged through
        # L2 (csrc/bilstm.hip) -- instead of 25 x (grouped skinny GEMM + pointwise launch).  The kernel takes 64 rows; a larger
        # batch runs it once per 64-row chunk (128 clips: 0.74 ms against 1.04 ms step by step, tools/bilstm_bench.py 128)
        for b0 in range(0, B, BILSTM_ROWS):
            b1 = min(B, b0 + BILSTM_ROWS)
            ops.bilstm_fwd([x.view(B, T, 4 * H)[b0:b1].reshape(-1, 4 * H) for x in xg], Whh, bih, bhh, out[b0:b1],
                           [h_[b0:b1] for h_ in hprev], [c_[b0:b1] for c_ in cst], [g_[b0:b1] for g_ in gates])
    else:
        _bilstm_steps_fwd(ops, xg, Whh, bih, bhh, out, hprev, cst, gates, B, T, H, ref)
    out2 = out.view(B * T, 2 * H)
    pd = m.p_drop if training else 0.0
    s.update(e=e, out=out2, hprev=hprev, cst=cst, gates=gates, pd=pd)
    ln = m.layernorm_lstm
    if m.baseline:
        hl = _empty(ref, B * T, 2 * H); st_l = _empty(ref, B * T, 2)
        ops.rowln_fwd(out2, ln.weight, ln.bias, hl, st_l, p1=pd, site1=SITE_LSTM, seed=seed)
        res = _empty(ref, B * T, H)
        lin(ops, hl, m.out_try.weight, res, m.out_try.bias)
        s.update(hl=hl, st_l=st_l)
        return res
    sa = m.self_attention
    D2 = 2 * H
    pe = sa.pe.pe[0, :T]
    x = _empty(ref, B * T, D2); st_l = _empty(ref, B * T, 2)
    ops.rowln_fwd(out2, ln.weight, ln.bias, x, st_l, pe=pe, p1=pd, site1=SITE_LSTM, p2=0.2 if training else 0.0,
                  site2=SITE_PE, seed=seed)
    Kp, Qp, Vp = _empty(ref, B * T, D2), _empty(ref, B * T, D2), _empty(ref, B * T, D2)
    # (riders as deep as the three projections join their launch: a fourth group of the stream-K launch instead of 82 us alone)
    ride = [g for g in extra if g[0].shape[1] == D2]
    extra = [g for g in extra if g[0].shape[1] != D2]
    ops.gemm(GEMM_NT, [(x, sa.K.weight, Kp), (x, sa.Q.weight, Qp), (x, sa.V.weight, Vp)] + ride)
    if extra:
        ops.gemm(GEMM_NT, extra)
    scale = 1.0 / math.sqrt(sa.attention_size)
    w = _empty(ref, B, T, T)
    att = _empty(ref, B * T, D2)
    if ops.sa_core_supported(T, D2):
        ops.sa_core_fwd(Kp.view(B, T, D2), Qp.view(B, T, D2), Vp.view(B, T, D2), w, att.view(B, T, D2), scale)
    else:
        lg = _empty(ref, B, T, T)
        ops.gemm(GEMM_NT, [(Kp.view(B, T, D2), Qp.view(B, T, D2), lg)], alpha=scale)
        ops.softmax_fwd(lg, w, B * T, T, 1)
        ops.gemm(GEMM_NN, [(w, Vp.view(B, T, D2), att.view(B, T, D2))])
    so = _empty(ref, B * T, H)
    lin(ops, att, sa.output_layer[0].weight, so)
    psa = sa.dropout if training else 0.0
    if psa > 0:
        ops.dropout(so, so, psa, seed, SITE_SA)
    res = _empty(ref, B * T, H); st_s = _empty(ref, B * T, 2)
    ops.rowln_fwd(so, m.layernorm_sa.weight, m.layernorm_sa.bias, res, st_s)
    s.update(x=x, st_l=st_l, Kp=Kp, Qp=Qp, Vp=Vp, w=w, att=att, so=so, st_s=st_s, scale=scale, psa=psa, pe=pe)
    return res


def _bilstm_steps_bwd(ops, gates, cst, dout3, Whh, dG, B, T, H, ref):
    """BiLSTM backward through time step by step: per step one pointwise launch + one grouped skinny NN GEMM (both directions)"""
    dcrec = [_empty(ref, B, H), _empty(ref, B, H)]
    ns_r = max(1, min(_nsplit_for(B, 2 * H, 1), 8))
    rb = _bwd_bounds(4 * H, B, ns_r)
    slabs = None                       # (S, B, 2H): slabs of [d h_prev of the forward dir | of the reverse dir]
    for step in range(T - 1, -1, -1):
        tt = [step, T - 1 - step]
        tp = [step - 1, T - step]
        last = step == T - 1
        # both directions in one launch; the recurrent gradient is read straight from the GEMM's slabs
        ops.lstm_pw_bwd_multi([dict(gates=gates[d][:, tt[d]], c=cst[d][:, tt[d]], dgates=dG[d][:, tt[d]], B=B, H=H,
                                    c_prev=cst[d][:, tp[d]] if step > 0 else None,
                                    dh=dout3[:, tt[d], d * H:(d + 1) * H],
                                    dh4=None if last else slabs[:, :, d * H:(d + 1) * H],
                                    dc_next=None if last else dcrec[d], dc_prev=dcrec[d]) for d in range(2)])
        if step > 0:
            # recurrent gradient of both directions: one grouped launch
            slabs = _empty(ref, len(rb), B, 2 * H)
            groups = []
            for d in range(2):
                dg = dG[d][:, tt[d]]
                for i, (k0, k1) in enumerate(rb):
                    groups.append((dg[:, k0:k1], Whh[d][k0:k1, :], slabs[i][:, d * H:(d + 1) * H]))
            ops.gemm(GEMM_NN, groups)


def encvis_bwd(ops, m, pfx, frames2d, B, T, sv, G, dres, training, seed):
    H = m.hidden_size
    ref = frames2d
    s = sv[pfx]
    name = pfx
    ln = m.layernorm_lstm
    out2 = s['out']
    D2 = 2 * H
    dout = _empty(ref, B * T, D2)
    if m.baseline:
        tn_grouped(ops, [(dres, s['hl'], G[name + '.out_try.weight'])], sv.get('tn_defer'))
        ops.colsum(dres, G[name + '.out_try.bias'], accum=True)
        dhl = _empty(ref, B * T, D2)
        ops.gemm(GEMM_NN, [(dres, m.out_try.weight, dhl)])
        nb = ops.rowln_bwd_nblk(B * T)
        part = _empty(ref, nb, 2, D2)
        ops.rowln_bwd(dhl, out2, ln.weight, ln.bias, dout, stats=s['st_l'], p1=s['pd'], site1=SITE_LSTM, seed=seed,
                      dgb_part=part)
        ln_grads(ops, part, G, name + '.layernorm_lstm', D2)
    else:
        sa = m.self_attention
        nb = ops.rowln_bwd_nblk(B * T)
        part = _empty(ref, nb, 2, H)
        dso = _empty(ref, B * T, H)
        ops.rowln_bwd(dres, s['so'], m.layernorm_sa.weight, m.layernorm_sa.bias, dso, stats=s['st_s'], dgb_part=part)
        ln_grads(ops, part, G, name + '.layernorm_sa', H)
        if s['psa'] > 0:
            ops.dropout(dso, dso, s['psa'], seed, SITE_SA)
        att, w, Kp, Qp, Vp, x, scale = s['att'], s['w'], s['Kp'], s['Qp'], s['Vp'], s['x'], s['scale']
        tn_grouped(ops, [(dso, att, G[name + '.self_attention.output_layer.0.weight'])], sv.get('tn_defer'))
        datt = _empty(ref, B * T, D2)
        ops.gemm(GEMM_NN, [(dso, sa.output_layer[0].weight, datt)])
        datt3 = datt.view(B, T, D2)
        dK = _empty(ref, B * T, D2); dQ = _empty(ref, B * T, D2); dV = _empty(ref, B * T, D2)
        if ops.sa_core_supported(T, D2):
            ops.sa_core_bwd(w, Kp.view(B, T, D2), Qp.view(B, T, D2), Vp.view(B, T, D2), datt3, dK.view(B, T, D2),
                            dQ.view(B, T, D2), dV.view(B, T, D2), scale)
        else:
            dw = _empty(ref, B, T, T)
            ops.gemm(GEMM_NT, [(datt3, Vp.view(B, T, D2), dw)])
            ops.gemm(GEMM_TN, [(w, datt3, dV.view(B, T, D2))])
            dlg = _empty(ref, B, T, T)
            ops.softmax_bwd(w, dw, dlg, B * T, T, 1)
            ops.gemm(GEMM_NN, [(dlg, Qp.view(B, T, D2), dK.view(B, T, D2))], alpha=scale)
            ops.gemm(GEMM_TN, [(dlg, Kp.view(B, T, D2), dQ.view(B, T, D2))], alpha=scale)
        dx = _empty(ref, B * T, D2)
        # dx = dK W_K + dQ W_Q + dV W_V: the three products as groups of one launch into slabs (each alone is 832 tiles on
        # 768 slots: a second round for 8 % of the work), then one fold
        sl = _empty(ref, 3, B * T, D2)
        ops.gemm(GEMM_NN, [(dK, sa.K.weight, sl[0]), (dQ, sa.Q.weight, sl[1]), (dV, sa.V.weight, sl[2])])
        ops.slab_reduce(sl, dx)
        tn_grouped(ops, [(dK, x, G[name + '.self_attention.K.weight']), (dQ, x, G[name + '.self_attention.Q.weight']),
                         (dV, x, G[name + '.self_attention.V.weight'])], sv.get('tn_defer'))
        nb = ops.rowln_bwd_nblk(B * T)
        part = _empty(ref, nb, 2, D2)
        ops.rowln_bwd(dx, out2, ln.weight, ln.bias, dout, stats=s['st_l'], pe=s['pe'], p1=s['pd'], site1=SITE_LSTM,
                      p2=0.2 if training else 0.0, site2=SITE_PE, seed=seed, dgb_part=part)
        ln_grads(ops, part, G, name + '.layernorm_lstm', D2)
    # ---- BiLSTM backward through time
    lstm = m.lstm
    Wih = [lstm.weight_ih_l0, lstm.weight_ih_l0_reverse]
    Whh = [lstm.weight_hh_l0, lstm.weight_hh_l0_reverse]
    sfx = ['', '_reverse']
    dout3 = dout.view(B, T, D2)
    gates, cst, hprev = s['gates'], s['cst'], s['hprev']
    dG = [_empty(ref, B, T, 4 * H), _empty(ref, B, T, 4 * H)]
    if getattr(ops, 'bilstm_supported', None) is not None and getattr(ops, 'persistent_bilstm_bwd', True) and ops.bilstm_supported(B, T, H):
        # all steps of both directions: one persistent launch (beyond 64 rows the chunked form only ties with the per-step
        # schedule -- 0.997 against 0.990 ms at 128 clips -- so that case stays step by step)
        ops.bilstm_bwd(gates, cst, dout3, Whh, dG)
    else:
        _bilstm_steps_bwd(ops, gates, cst, dout3, Whh, dG, B, T, H, ref)
    de = _empty(ref, B * T, H)
    e = s['e']
    tn_grouped(ops, [(dG[d].view(B * T, 4 * H), src, G[name + '.lstm.' + wn + sfx[d]])
                     for d in range(2) for src, wn in ((e, 'weight_ih_l0'), (hprev[d].view(B * T, H), 'weight_hh_l0'))],
               sv.get('tn_defer'))
    # de = sum_d dG_d W_ih_d: both directions in ONE launch, each contraction (4H deep for a 416-tile output) in two halves,
    # four slabs folded once (two accumulating launches before: 2 x 161 us; a single product K-split four ways: 146 against 168)
    dgs = [dG[d].view(B * T, 4 * H) for d in range(2)]
    kh = (4 * H) // 2
    if kh % 32 == 0:
        sl = _empty(ref, 4, B * T, H)
        ops.gemm(GEMM_NN, [(dgs[d][:, i * kh:(i + 1) * kh], Wih[d][i * kh:(i + 1) * kh, :], sl[2 * d + i]) for d in range(2) for i in range(2)])
        ops.slab_reduce(sl, de)
    else:
        for d in range(2):
            ops.gemm(GEMM_NN, [(dgs[d], Wih[d], de)], flags=F_ACCUM if d else 0)
    for d in range(2):
        ops.colsum2(dgs[d], G[name + '.lstm.bias_ih_l0' + sfx[d]], G[name + '.lstm.bias_hh_l0' + sfx[d]], accum=True)
    tn_grouped(ops, [(de, frames2d, G[name + '.linear_embed.weight'])], sv.get('tn_defer'))
    ops.colsum(de, G[name + '.linear_embed.bias'], accum=True)


# ================================================================================================ decoder
class DecPlan(object):
    """Column layout of the two LSTMCells' input weights (models/layer.py:313-327,571,587)."""

    def __init__(self, dec, n_global):
        self.H, self.W = dec.visual_hidden_size, dec.word_size
        self.Q, self.D = dec.query_hidden_size, dec.decode_hidden_size
        self.G = n_global                      # width of the global feature (2H multi-modal, H baseline)
        self.ns = 2 if dec.multi_modal else 1  # attention streams feeding the language LSTM
        # query_lstm input = [lang_h (D) | global (G) | word (W)]
        self.q_lang = (0, self.D)
        self.q_glob = (self.D, self.D + self.G)
        self.q_word = (self.D + self.G, self.D + self.G + self.W)
        # lang_lstm input = [ctx_1 (H) | (ctx_2 (H)) | q_cur (Q)]
        self.l_ctx = [(i * self.H, (i + 1) * self.H) for i in range(self.ns)]
        self.l_q = (self.ns * self.H, self.ns * self.H + self.Q)


def _att_modules(dec):
    return [dec.context_att] + ([dec.context_att_2] if dec.multi_modal else [])


def dec_prepare(ops, dec, mems, sv, training, seed):
    """Step-invariant work: global feature + its gate contribution, K' and V' of every attention stream.
    mems: list of proposal tensors (B,P,H) -- one per attention stream (already concatenated for the
    non-multi-modal two-stream case).  gsrc: tensors averaged into the global feature."""
    s = sv['dec'] = {}
    gsrc = sv['dec_gsrc']
    ref = mems[0]
    B = ref.shape[0]
    H = dec.visual_hidden_size
    plan = DecPlan(dec, H * len(gsrc))
    Q = plan.Q
    if sv.get('step_feats') is not None:
        gfeat = sv['step_feats']               # models/layer.py:404-405: a given global feature replaces the proposals' means
        assert gfeat.shape == (B, plan.G), (tuple(gfeat.shape), (B, plan.G))
    else:
        gfeat = _empty(ref, B, plan.G)
        for i, g in enumerate(gsrc):
            ops.mean_rows_fwd(g, gfeat[:, i * H:(i + 1) * H])
    Wq = dec.query_lstm.weight_ih
    gq = _empty(ref, B, 4 * Q)
    lin(ops, gfeat, Wq[:, plan.q_glob[0]:plan.q_glob[1]], gq)
    atts = _att_modules(dec)
    m2s = [mem.reshape(mem.shape[0] * mem.shape[1], H) for mem in mems]
    Kc = [_empty(ref, m2.shape[0], H) for m2 in m2s]
    Vc = [_empty(ref, m2.shape[0], H) for m2 in m2s]
    kps = [_empty(ref, m2.shape[0], Q) for m2 in m2s]
    vps = [_empty(ref, m2.shape[0], H) for m2 in m2s]
    # both streams per launch: K = m W_K^T, V = m W_V^T;  K' = K W_Q;  V' = V W_O^T
    gemm_bucketed(ops, GEMM_NT, [g for att, m2, K, V in zip(atts, m2s, Kc, Vc) for g in ((m2, att.K.weight, K), (m2, att.V.weight, V))])
    gemm_bucketed(ops, GEMM_NN, [(K, att.Q.weight, kp) for att, K, kp in zip(atts, Kc, kps)])
    gemm_bucketed(ops, GEMM_NT, [(V, att.output_layer[0].weight, vp) for att, V, vp in zip(atts, Vc, vps)])
    Kp = [kp.view(mem.shape[0], mem.shape[1], Q) for kp, mem in zip(kps, mems)]
    Vp = [vp.view(mem.shape[0], mem.shape[1], H) for vp, mem in zip(vps, mems)]
    s.update(plan=plan, gfeat=gfeat, gq=gq, K=Kc, V=Vc, Kp=Kp, Vp=Vp, mems=mems)
    return s


def dec_step(ops, dec, s, t, ref, training, seed, B, word_dropout=True, sample=None):
    """One Decoder.decode (models/layer.py:569-602) on time-major state buffers; step t reads slot t, writes t+1."""
    plan = s['plan']
    H, Q, D = plan.H, plan.Q, plan.D
    ql, ll = dec.query_lstm, dec.lang_lstm
    pd = dec.p_drop if training else 0.0
    site = STEP_SITE * (t + 1)
    # ---- query LSTM
    segs = [(s['WE'][t], ql.weight_ih[:, plan.q_word[0]:plan.q_word[1]])]
    if t > 0 or s.get('warm', False):
        segs += [(s['LHP'][t], ql.weight_ih[:, plan.q_lang[0]:plan.q_lang[1]]), (s['QH'][t], ql.weight_hh)]
    slabs = seg_gemm_nt(ops, segs, B, 4 * Q, ref)
    # ---- cell pointwise -> LN -> attention over the cached proposals -> tanh -> LN: one launch per step
    ns = plan.ns
    lnq = dec.query_lstm_layernorm
    atts = _att_modules(dec)
    ops.dec_mid_fwd(slabs, s['gq'], ql.bias_ih, ql.bias_hh, s['QC'][t], s['QC'][t + 1], s['QH'][t + 1], s['GQ'][t],
                    (lnq.weight, lnq.bias), s['QCUR'][t], s['ST_Q'][t], pd, site + SITE_QUERY, s['Kp'], s['Vp'],
                    [(a.output_layer[2].weight, a.output_layer[2].bias) for a in atts],
                    [s['CPRE'][i][t] for i in range(ns)], [s['CTX'][i][t] for i in range(ns)],
                    [s['ST_C'][i][t] for i in range(ns)], s['ALPHA'][t],
                    [a.dropout if training else 0.0 for a in atts], [site + SITE_ATT1 + i for i in range(ns)],
                    1.0 / math.sqrt(H), seed=seed, **({'kv_div': s['kv_div']} if s.get('kv_div', 1) > 1 else {}))
    # ---- language LSTM
    segs = [(s['CTX'][i][t], ll.weight_ih[:, plan.l_ctx[i][0]:plan.l_ctx[i][1]]) for i in range(ns)]
    segs.append((s['QCUR'][t], ll.weight_ih[:, plan.l_q[0]:plan.l_q[1]]))
    if t > 0 or s.get('warm', False):
        segs.append((s['LHP'][t], ll.weight_hh))
    slabs = seg_gemm_nt(ops, segs, B, 4 * D, ref)
    lnl = dec.lang_lstm_layernorm
    ops.dec_tail_fwd(slabs, ll.bias_ih, ll.bias_hh, s['LC'][t], s['LC'][t + 1], s['LHP'][t + 1], s['GL'][t],
                     (lnl.weight, lnl.bias), s['DOUT'][t], s['ST_L'][t], pd, site + SITE_LANG, seed=seed,
                     **({'sample': sample} if sample is not None else {}))


def plan_D(s):
    return s['plan'].D


def dec_logits(ops, dec, s, t0, t1, skip_if=None):
    """logits of steps [t0,t1): word_restore of tanh(LN(lang_h)) (models/layer.py:599-600); dec_step left the LN output in DOUT.
    skip_if: device flag -- the launch is a no-op when it is set (teacher-forced step of a replayed graph)."""
    plan = s['plan']
    D = plan.D
    B = s['LHP'].shape[1]
    V = dec.vocab_size
    n = (t1 - t0) * B
    lin(ops, s['DOUT'][t0:t1].view(n, D), dec.word_restore.weight, s['LOGITS'][t0:t1].view(n, V), dec.word_restore.bias,
        skip_if=skip_if)


def dec_alloc(dec, s, ref, B, L):
    plan = s['plan']
    H, W, Q, D, ns = plan.H, plan.W, plan.Q, plan.D, plan.ns
    V = dec.vocab_size
    P = s['Kp'][0].shape[1]
    # the four recurrent state arrays start at zero: ONE fill (each torch.zeros is a 5-us launch in the replayed step)
    zst = _zeros(ref, (L + 1) * B * (2 * D + 2 * Q))
    o = 0
    for key, wd in (('LHP', D), ('QH', Q), ('QC', Q), ('LC', D)):
        s[key] = zst[o:o + (L + 1) * B * wd].view(L + 1, B, wd)
        o += (L + 1) * B * wd
    s['WE'] = _empty(ref, L + 1, B, W)
    s['IDS'] = _zeros(ref, L + 1, B, dtype=torch.int64)
    s['QCUR'] = _empty(ref, L, B, Q)
    s['ST_Q'] = _empty(ref, L, B, 2)
    s['CPRE'] = [_empty(ref, L, B, H) for _ in range(ns)]
    s['CTX'] = [_empty(ref, L, B, H) for _ in range(ns)]
    s['ST_C'] = [_empty(ref, L, B, 2) for _ in range(ns)]
    s['GQ'] = _empty(ref, L, B, 4 * Q)
    s['GL'] = _empty(ref, L, B, 4 * D)
    s['ALPHA'] = _empty(ref, L, B, ns * P)
    s['DOUT'] = _empty(ref, L, B, D)
    s['ST_L'] = _empty(ref, L, B, 2)
    s['LOGITS'] = _empty(ref, L, B, V)


def dec_fwd(ops, dec, mems, sv, captions, L, coins, training, seed, dev_coins=None):
    """Decoder.forward, training / greedy branch (models/layer.py:394-447).
    coins[i] True -> step i feeds captions[:, i] to step i+1, else the argmax of its own logits.
    captions None -> greedy inference (all coins False).  Returns the decoder state dict.
    dev_coins: optional int32 device array of the same coins; then the word choice happens on device
    (`select_embed`) and the launch sequence no longer depends on the coin pattern (hipGraph capture)."""
    s = dec_prepare(ops, dec, mems, sv, training, seed)
    ref = mems[0]
    B = ref.shape[0]
    dec_alloc(dec, s, ref, B, L)
    E = dec.word_embed.weight
    pw = dec.p_drop if training else 0.0
    ids = s['IDS']
    ids[0].fill_(dec.vocab('<start>'))
    s['pw'] = pw
    if dev_coins is not None:
        ops.embed_fwd(E, ids[0], s['WE'][0], p=pw, seed=seed, site=SITE_WORD, row0=0)
        pre = captions is not None and L > 1
        if pre:
            # the teacher-forced choice of EVERY step in one launch (ids[t + 1] = captions[:, t], its embedding under the mask rows
            # of slot t + 1): the per-step select_embed then only has work on sampled steps -- with the reference's schedule
            # (utils ss epsilon 0.95 .. ) that is about one step in twenty
            ids[1:L].copy_(captions[:, :L - 1].t())
            ops.embed_fwd(E, ids[1:L].reshape(-1), s['WE'][1:L].view((L - 1) * B, -1), p=pw, seed=seed, site=SITE_WORD, row0=B)
        # small vocabularies: the language cell's launch samples the next word itself on the steps whose coin says so (every
        # workgroup projects its own row) -- instead of a vocabulary projection and a select launch per step that do nothing
        # on the ~19 teacher-forced steps out of 20 but cost a launch each
        fused_sample = pre and getattr(ops, 'dec_tail_sample_supported', None) is not None and \
            ops.dec_tail_sample_supported(dec.vocab_size, plan_D(s))
        for t in range(L):
            if fused_sample and t + 1 < L:
                dec_step(ops, dec, s, t, ref, training, seed, B,
                         sample=dict(coins=dev_coins, t=t, W=dec.word_restore.weight, b=dec.word_restore.bias, E=E, ids_out=ids[t + 1],
                                     we_out=s['WE'][t + 1], p=pw, site=SITE_WORD, row0=(t + 1) * B))
                continue
            dec_step(ops, dec, s, t, ref, training, seed, B)
            if t + 1 < L:
                # the step's logits are needed now only if the next word is sampled from them: on a teacher-forced step
                # (device coin set) the per-step vocab projection is a no-op and select_embed takes the caption word
                dec_logits(ops, dec, s, t, t + 1, skip_if=dev_coins[t:t + 1])
                ops.select_embed(s['LOGITS'][t], captions, t, dev_coins, E, ids[t + 1], s['WE'][t + 1], p=pw, seed=seed,
                                 site=SITE_WORD, row0=(t + 1) * B, prefilled=pre)
        dec_logits(ops, dec, s, 0, L)          # logits of all steps for the loss, one (L*B)-row product
        return s
    if captions is not None:
        tf_steps = [i for i in range(L) if coins[i]]
        for i in tf_steps:
            ids[i + 1].copy_(captions[:, i])
    # embed every step input that is known up front (start token + teacher-forced words), word dropout per row
    ops.embed_fwd(E, ids.view(-1), s['WE'].view((L + 1) * B, -1), p=pw, seed=seed, site=SITE_WORD)
    for t in range(L):
        dec_step(ops, dec, s, t, ref, training, seed, B)
        if not coins[t]:
            dec_logits(ops, dec, s, t, t + 1)
            ops.argmax(s['LOGITS'][t], ids[t + 1])
            # same (seed, site, row) mask stream as the bulk call above: rows (t+1)*B.. of the WE matrix
            ops.embed_fwd(E, ids[t + 1], s['WE'][t + 1], p=pw, seed=seed, site=SITE_WORD, row0=(t + 1) * B)
    if any(coins[t] for t in range(L)):
        dec_logits(ops, dec, s, 0, L)          # (greedy inference: every step's logits were already written above)
    return s


def dec_bwd(ops, dec, sv, G, dlogits_tm, seed, training, dalpha_tm=None):
    """Backward of dec_fwd.  dlogits_tm (L,B,V).  Returns list of d(mem) per attention stream and d(global src)."""
    s = sv['dec']
    plan = s['plan']
    H, W, Q, D, ns = plan.H, plan.W, plan.Q, plan.D, plan.ns
    L, B, V = dlogits_tm.shape
    ref = dlogits_tm
    ql, ll = dec.query_lstm, dec.lang_lstm
    pd = dec.p_drop if training else 0.0
    n = L * B
    dl2 = dlogits_tm.view(n, V)
    # ---- vocab projection + tanh(LN(lang_h))
    tn_grouped(ops, [(dl2, s['DOUT'].view(n, D), G['decoder.word_restore.weight'])], sv.get('tn_defer'))
    ops.colsum(dl2, G['decoder.word_restore.bias'], accum=True)
    ddout = _empty(ref, n, D)
    k32 = V // 32 * 32
    if V - k32 and k32 >= 4096 and getattr(ops, 'stream_k', False):
        # a 10 000-word vocabulary: the stream-K kernel takes contractions in whole 32-deep stages, so the product runs on it up to
        # 9 984 and the last 16 words are added by a second, tiny launch (553 us on the small tiles against 420 + 15)
        ops.gemm(GEMM_NN, [(dl2[:, :k32], dec.word_restore.weight[:k32], ddout)])
        ops.gemm(GEMM_NN, [(dl2[:, k32:], dec.word_restore.weight[k32:], ddout)], flags=F_ACCUM)
    else:
        ops.gemm(GEMM_NN, [(dl2, dec.word_restore.weight, ddout)])
    lnl = dec.lang_lstm_layernorm
    nb = ops.rowln_bwd_nblk(n)
    part = _empty(ref, nb, 2, D)
    dLHo = _empty(ref, L, B, D)
    ops.rowln_bwd(ddout, s['LHP'][1:].view(n, D), lnl.weight, lnl.bias, dLHo.view(n, D), stats=s['ST_L'].view(n, 2),
                  post_tanh=1, dgb_part=part)
    ln_grads(ops, part, G, 'decoder.lang_lstm_layernorm', D)

    dGQ = _empty(ref, L, B, 4 * Q)
    dGL = _empty(ref, L, B, 4 * D)
    dQC = _zeros(ref, B, Q)
    dLC = _zeros(ref, B, D)
    dLHrec = _empty(ref, B, D)           # recurrent part of d lang_h, written by step t+1's fused kernel
    P = s['Kp'][0].shape[1]
    dCPRE = [_empty(ref, L, B, H) for _ in range(ns)]       # kept per step: contracted into dV' after the loop
    dS = _empty(ref, L, B, ns * P)                           # score grads per step: contracted into dK'
    atts = _att_modules(dec)
    part_q = _empty(ref, L, B, 2, Q)
    part_c = [_empty(ref, L, B, 2, H) for _ in range(ns)]
    lnq = dec.query_lstm_layernorm
    lnc_g = [att.output_layer[2].weight for att in atts]
    p_att = [att.dropout if training else 0.0 for att in atts]
    scale = 1.0 / math.sqrt(H)
    Wl_in = ll.weight_ih
    Wq_rec = [ql.weight_hh, ql.weight_ih[:, plan.q_lang[0]:plan.q_lang[1]]]
    wx = ns * H + Q + D
    ql_slabs = None                       # slabs of [d query_h(recurrent) | d lang_h(query-input part)] from step t+1
    for t in range(L - 1, -1, -1):
        site = STEP_SITE * (t + 1)
        # language cell: total grad on lang h_t (dropped) = vocab head part + (step t+1) lang-recurrent part +
        # query-input part; the three are summed inside the cell kernel (they share the dropout mask of h_t)
        rec = t < L - 1
        ops.lstm_pw_bwd(s['GL'][t], s['LC'][t + 1], dGL[t], B, D, c_prev=s['LC'][t], dh2=dLHo[t],
                        dh3=dLHrec if rec else None, dh4=ql_slabs[:, :, Q:] if rec else None, dc_next=dLC, dc_prev=dLC,
                        p=pd, site=site + SITE_LANG, seed=seed)
        # input grads of the language cell: slabs of dGL[t] . [W_ih | W_hh] from one grouped launch ...
        xl_slabs = gemm_nn_multi_slabs(ops, dGL[t], [Wl_in, ll.weight_hh] if t > 0 else [Wl_in], wx, ref)
        # ... consumed by ONE launch: slab sum -> output_layer LN bwd -> attention bwd -> query LN bwd -> query cell bwd
        ops.dec_mid_bwd(xl_slabs, dLHrec if t > 0 else None, [s['CPRE'][i][t] for i in range(ns)],
                        [s['ST_C'][i][t] for i in range(ns)], lnc_g, [part_c[i][t] for i in range(ns)],
                        [dCPRE[i][t] for i in range(ns)], p_att, [site + SITE_ATT1 + i for i in range(ns)], s['Kp'], s['Vp'],
                        s['ALPHA'][t], dalpha_tm[t] if dalpha_tm is not None else None, dS[t], s['QH'][t + 1], s['ST_Q'][t],
                        lnq.weight, part_q[t], pd, site + SITE_QUERY, ql_slabs[:, :, :Q] if rec else None, s['GQ'][t],
                        s['QC'][t + 1], s['QC'][t], dQC, dGQ[t], scale, seed=seed)
        if t > 0:
            ql_slabs = gemm_nn_multi_slabs(ops, dGQ[t], Wq_rec, Q + D, ref)
    dKp = [torch.empty_like(k) for k in s['Kp']]
    dVp = [torch.empty_like(v) for v in s['Vp']]
    ops.decatt_cache_grads(s['ALPHA'], dS, s['QCUR'], dCPRE, dKp, dVp)
    # ---- LayerNorm parameter grads of the per-step norms
    ln_grads(ops, part_q.view(L * B, 2, Q), G, 'decoder.query_lstm_layernorm', Q)
    att_names = ['decoder.context_att', 'decoder.context_att_2']
    for i in range(ns):
        ln_grads(ops, part_c[i].view(L * B, 2, H), G, att_names[i] + '.output_layer.2', H)
    # ---- weight gradients: one (L*B)-deep TN GEMM per weight block
    dgq2, dgl2 = dGQ.view(n, 4 * Q), dGL.view(n, 4 * D)
    Gq_ih, Gl_ih = G['decoder.query_lstm.weight_ih'], G['decoder.lang_lstm.weight_ih']
    dgq_sum = _empty(ref, B, 4 * Q)
    ops.slab_reduce(dGQ, dgq_sum)
    lhp, qh = s['LHP'][:L].view(n, D), s['QH'][:L].view(n, Q)
    tn_grouped(ops, [(dgq2, lhp, Gq_ih[:, plan.q_lang[0]:plan.q_lang[1]]),
                     (dgq2, s['WE'][:L].view(n, W), Gq_ih[:, plan.q_word[0]:plan.q_word[1]]),
                     (dgq2, qh, G['decoder.query_lstm.weight_hh'])] +
               [(dgl2, s['CTX'][i].view(n, H), Gl_ih[:, plan.l_ctx[i][0]:plan.l_ctx[i][1]]) for i in range(ns)] +
               [(dgl2, s['QCUR'].view(n, Q), Gl_ih[:, plan.l_q[0]:plan.l_q[1]]),
                (dgl2, lhp, G['decoder.lang_lstm.weight_hh'])], sv.get('tn_defer'))
    gq_glob = Gq_ih[:, plan.q_glob[0]:plan.q_glob[1]]
    ops.gemm(GEMM_TN, [(dgq_sum, s['gfeat'], gq_glob)], flags=_accum_always(ops, gq_glob))
    ops.colsum2(dgq2, G['decoder.query_lstm.bias_ih'], G['decoder.query_lstm.bias_hh'], accum=True)
    ops.colsum2(dgl2, G['decoder.lang_lstm.bias_ih'], G['decoder.lang_lstm.bias_hh'], accum=True)
    # ---- word embedding rows
    dWE = _empty(ref, n, W)
    Wword = ql.weight_ih[:, plan.q_word[0]:plan.q_word[1]]
    if n >= 512 and 4 * Q >= 2048:
        # (L*B x 4Q) @ (4Q x 300): 26 x 5 output tiles for a 4096-deep contraction -- 130 workgroups for 110 us; the
        # contraction in 1024-deep slabs puts 4x as many on the chip
        kb = [(k, min(4 * Q, k + 1024)) for k in range(0, 4 * Q, 1024)]
        sl = _empty(ref, len(kb), n, W)
        ops.gemm(GEMM_NN, [(dgq2[:, k0:k1], Wword[k0:k1], sl[i]) for i, (k0, k1) in enumerate(kb)])
        ops.slab_reduce(sl, dWE)
    else:
        ops.gemm(GEMM_NN, [(dgq2, Wword, dWE)])
    # one mask stream (site SITE_WORD, row = slot*B + b) covers the bulk embed and the argmax re-embeds
    ops.embed_bwd(dWE, s['IDS'][:L].view(-1), G['decoder.word_embed.weight'], p=s['pw'], seed=seed, site=SITE_WORD)
    # ---- global feature
    dgfeat = _empty(ref, B, plan.G)
    gemm_nn_split(ops, dgq_sum, ql.weight_ih[:, plan.q_glob[0]:plan.q_glob[1]], dgfeat, ref)
    # ---- attention caches: K' = K W_Q, V' = V W_O^T, K = m W_K^T, V = m W_V^T
    m2s = [mem.reshape(mem.shape[0] * mem.shape[1], H) for mem in s['mems']]
    dkps = [dKp[i].view(m2s[i].shape[0], Q) for i in range(ns)]
    dvps = [dVp[i].view(m2s[i].shape[0], H) for i in range(ns)]
    dKs = [_empty(ref, m2.shape[0], H) for m2 in m2s]
    dVs = [_empty(ref, m2.shape[0], H) for m2 in m2s]
    dms = [_empty(ref, m2.shape[0], H) for m2 in m2s]
    R = range(ns)
    # every product once for all streams: dW_Q = K^T dK', dW_O = dV'^T V | dK = dK' W_Q^T | dV = dV' W_O |
    # dW_K = dK^T m, dW_V = dV^T m | dm = dK W_K (+ dV W_V)
    tn_grouped(ops, [g for i in R for g in ((s['K'][i], dkps[i], G[att_names[i] + '.Q.weight']),
                                            (dvps[i], s['V'][i], G[att_names[i] + '.output_layer.0.weight']))], sv.get('tn_defer'))
    gemm_bucketed(ops, GEMM_NT, [(dkps[i], atts[i].Q.weight, dKs[i]) for i in R])
    gemm_bucketed(ops, GEMM_NN, [(dvps[i], atts[i].output_layer[0].weight, dVs[i]) for i in R])
    tn_grouped(ops, [g for i in R for g in ((dKs[i], m2s[i], G[att_names[i] + '.K.weight']),
                                            (dVs[i], m2s[i], G[att_names[i] + '.V.weight']))], sv.get('tn_defer'))
    gemm_bucketed(ops, GEMM_NN, [(dKs[i], atts[i].K.weight, dms[i]) for i in R])
    gemm_bucketed(ops, GEMM_NN, [(dVs[i], atts[i].V.weight, dms[i]) for i in R], flags=F_ACCUM)
    dmems = [dms[i].view(s['mems'][i].shape) for i in R]
    return dmems, dgfeat
