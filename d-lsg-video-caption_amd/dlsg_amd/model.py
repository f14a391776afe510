"""Drop-in model surface of the hot path: CapGnnModel / CapBaseline1 with the reference's construction and
forward() signatures (models/model.py:25-43,94-107), state_dict layout and
`.decoder.decode_tokens` / `update_beam_size`, running on the HIP kernels of libdlsg_hip.so.

`forward` is one torch.autograd.Function around the hand-scheduled engine, so reference-style callers
(`loss.backward()` + torch.optim, DDP wrapping) work unchanged; `Trainer` is the fast path the benchmark uses:
fused ragged cross-entropy, hand-scheduled backward into a flat gradient arena, bucketed RCCL all-reduce that
overlaps the encoder backward, and one fused Adam kernel over the flat parameter arena.
"""
import math
import os
import random

import torch
import torch.nn as nn

from . import engine as E
from .modules import CapGnnEncoder, Decoder, EncoderVisual

_ALIGN = 64  # floats: every parameter starts 256-B aligned inside the arena (vector loads need 16 B)


def _default_ops():
    from .hip import HipOps
    return HipOps()


class _ArenaModule(nn.Module):
    """A module whose parameters are views of ONE flat fp32 arena (plus a gradient arena of the same layout) and that launches
    kernels through an `ops` handle: shared by the generator models and the critic (gan.DiscV2)."""

    def __init__(self):
        super().__init__()
        self._ops_obj = None
        self._flat = None
        self._gflat = None
        self._offsets = None

    # ------------------------------------------------------------------ copy / pickle
    def __getstate__(self):
        """copy.deepcopy(model) / torch.save(model): the ctypes library handle and the `random` module cannot be pickled and
        the arenas are rebuilt on first use (flatten_parameters_ notices that the copied parameters are not its views)."""
        st = dict(self.__dict__)
        st['_ops_obj'] = None
        if 'rng' in st:
            st['rng'] = None
        for k in ('_flat', '_gflat', '_offsets', '_G'):
            st[k] = None
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        if 'rng' in st:
            self.rng = random

    # ------------------------------------------------------------------ kernels handle
    @property
    def ops(self):
        if self._ops_obj is None:
            self._ops_obj = _default_ops()   # raises when the HIP library / GPU is missing: no fallback
        return self._ops_obj

    def set_ops(self, ops):
        self._ops_obj = ops
        return self

    # ------------------------------------------------------------------ arenas
    def _arena_ok(self):
        if self._flat is None:
            return False
        for name, p in self.named_parameters():
            o = self._offsets[name]
            if p.device != self._flat.device or p.data_ptr() != self._flat.data_ptr() + 4 * o:
                return False
        return True

    def flatten_parameters_(self):
        """(Re)pack all parameters into one flat fp32 arena (views keep the nn.Parameter objects alive, so
        state_dict / load_state_dict / optimizers keep working) and allocate the matching gradient arena."""
        dec = getattr(self, 'decoder', None)
        if dec is not None and getattr(dec, '_owner', None) is None:
            import weakref
            object.__setattr__(dec, '_owner', weakref.ref(self))        # (a deep copy / unpickled model: see _HipModel.__setattr__)
        if self._arena_ok():
            return
        named = list(self.named_parameters())
        dev = named[0][1].device
        offsets, tot = {}, 0
        for name, p in named:
            offsets[name] = tot
            tot += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        flat = torch.zeros(tot, dtype=torch.float32, device=dev)
        for name, p in named:
            view = flat[offsets[name]:offsets[name] + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
        self._flat, self._offsets = flat, offsets
        self._gflat = torch.zeros(tot, dtype=torch.float32, device=dev)
        self._G = E.Grads(named, self._gflat, offsets)

    def grad_views(self):
        return self._G


class _HipModel(_ArenaModule):
    """Shared machinery of the generator models: arenas, ops handle, autograd bridge."""

    def __init__(self):
        super().__init__()
        self.rng = random          # scheduled-sampling coins come from Python's `random`, like layer.py:432
        self.seed_counter = 0
        self.fused_o2v = True
        # GEMM arithmetic: 'fp32' = exact fp32 MFMA everywhere; 'x3_bwd' = split-bf16 (3 bf16 MFMAs per product,
        # ~1e-5 relative error) for the backward products only; 'x3_all' = split-bf16 for forward and backward.
        self.gemm_precision = 'fp32'

    def __setattr__(self, name, value):
        super().__setattr__(name, value)
        if name == 'decoder' and isinstance(value, nn.Module):
            # Decoder.forward called on its own (models/layer.py:394) launches through the owning model's binding and arena
            import weakref
            object.__setattr__(value, '_owner', weakref.ref(self))

    @staticmethod
    def check_kernel_limits(args, attended_rows, what):
        """The fused decoder-step kernels (csrc/decstep.hip: MAXW, MAXP) hold one batch row's vectors in LDS: widths up to
        2048, at most 72 attended rows per attention stream (72 = the longest clip the reference's positional
        encoding admits, sublayer.py:87; the baseline decoders attend over the frame nodes).  Checked at construction, with names, instead of an EINVAL
        from the first launch."""
        for k in ('query_hidden_size', 'decode_hidden_size', 'visual_hidden_size'):
            if getattr(args, k) > 2048:
                raise ValueError('%s = %d: the fused decoder step supports widths up to 2048' % (k, getattr(args, k)))
        if attended_rows > 72:
            raise ValueError('%s = %d: the fused decoder step attends over at most 72 rows per stream' % (what, attended_rows))

    stream_k_in_backward = True        # False: the backward's products stay on the tiled kernels
    sk_backward_cu_budget = 0          # > 0 (a Trainer with several ranks sets it): workgroups of the backward's stream-K launches,
                                       # the other CUs are left to the gradient all-reduces that run beside the backward

    def _gemm_flags(self, backward):
        from .hip import F_BF16X3, F_NOSK
        mode = self.gemm_precision
        assert mode in ('fp32', 'x3_bwd', 'x3_all'), mode
        fl = F_BF16X3 if (mode == 'x3_all' or (mode == 'x3_bwd' and backward)) else 0
        if backward and not self.stream_k_in_backward:
            fl |= F_NOSK
        return fl

    def _gemm_policy(self, backward):
        """what every dlsg_gemm call of the pass that starts here carries: arithmetic + stream-K flags, the stream-K CU budget"""
        ops = self.ops
        ops.extra_flags = self._gemm_flags(backward)
        ops.sk_cu_budget = self.sk_backward_cu_budget if backward else 0
        if backward:
            ops.grad_written = set()        # engine._accum_flag: the first product into a gradient block stores, later ones add

    merge_weight_grads = True          # see engine.tn_grouped
    _defer_ok = True                   # weight gradients may be collected and launched grouped ...
    _flush_at_buckets = False          # ... at the end of the backward (one rank) or at every gradient bucket (set by a Trainer that reduces buckets)

    def next_seed(self):
        self.seed_counter += 1
        return (0x5DEECE66D * self.seed_counter + 0xB) & 0xFFFFFFFFFFFF

    # ------------------------------------------------------------------ to be provided by subclasses
    def _engine_forward(self, frames, regions, captions, L, coins, training, seed, sv):
        raise NotImplementedError

    def _engine_backward(self, sv, dlogits_tm, dobj, dmot, dalpha_tm, training, seed):
        raise NotImplementedError

    def _draw_coins(self, L, infer, tf_ratio):
        # reference draws one coin per step only when not inferring (layer.py:432)
        if infer:
            return [False] * L
        return [self.rng.random() < tf_ratio for _ in range(L)]


class _ModelFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, frames, regions, captions, L, coins, seed, *params):
        sv = {}
        training = model.training
        outs = model._engine_forward(frames, regions, captions, L, coins, training, seed, sv)
        ctx.model, ctx.sv, ctx.seed, ctx.training = model, sv, seed, training
        ctx.nparams = len(params)
        return outs

    @staticmethod
    def backward(ctx, dlogits, dobj, dmot, dalpha):
        model, ops = ctx.model, ctx.model.ops
        Bn, L, V = dlogits.shape
        dl_tm = torch.empty(L, Bn, V, dtype=torch.float32, device=dlogits.device)
        ops.permute_tb(dlogits.contiguous(), dl_tm)          # (B,L,V) -> (L,B,V)
        da_tm = None
        if dalpha is not None and dalpha.numel() > 0:
            da_tm = torch.empty(L, Bn, dalpha.shape[-1], dtype=torch.float32, device=dlogits.device)
            ops.permute_tb(dalpha.contiguous(), da_tm)
        model._engine_backward(ctx.sv, dl_tm, dobj, dmot, da_tm, ctx.training, ctx.seed)
        G = model.grad_views()
        grads = []
        for name, p in model.named_parameters():
            grads.append(G[name].clone() if (name not in model.unused_parameters and p.requires_grad) else None)
        ctx.sv = None
        return (None,) * 7 + tuple(grads)


class CapGnnModel(_HipModel):
    """models/model.py:25-43.  forward(visual_feats, region_feats, caption, max_words=None,
    teacher_forcing_ratio=1.0) -> (outputs, obj_proposals, motion_proposals, alpha_all)."""

    @property
    def unused_parameters(self):
        """constructed by the reference but never used in forward: they get no gradient;
        with num_obj < 5 the graph is skipped (layer.py:181-182) and obj_visual_norm is unused too."""
        names = ['encoder.obj_encoder.att_l2l_norm.weight', 'encoder.obj_encoder.att_l2l_norm.bias',
                 'encoder.motion_encoder.att_l2l_norm.weight', 'encoder.motion_encoder.att_l2l_norm.bias',
                 'decoder.context_layernorm.weight', 'decoder.context_layernorm.bias']
        if not self.encoder.obj_encoder.has_obj:
            for e in ('obj_encoder', 'motion_encoder'):
                names += ['encoder.%s.obj_visual_norm.1.weight' % e, 'encoder.%s.obj_visual_norm.1.bias' % e]
        return frozenset(names)

    def __init__(self, args, vocab):
        super().__init__()
        self.check_kernel_limits(args, args.num_proposals, 'num_proposals')
        self.use_visual_gan = args.use_visual_gan
        self.encoder = CapGnnEncoder(args)
        self.decoder = Decoder(args, vocab, multi_modal=True)

    def update_beam_size(self, beam_size):
        self.decoder.update_beam_size(beam_size)

    def load_encoder(self, model, model_path):
        """models/model.py:45-53: load `model_path` into `model`, graft its `.encoder` and `.decoder.word_embed` into this
        model and freeze the word embedding.  The grafted modules are shared objects, as in the reference; this model's
        arenas are re-packed around them on the next use (the donor is not meant to be used afterwards, run_gun.py keeps
        only the grafted model).  `Trainer` leaves parameters with `requires_grad == False` untouched."""
        if model_path is not None:
            model.load_state_dict(torch.load(model_path, map_location=next(self.parameters()).device))
        self.encoder = model.encoder
        self.decoder.word_embed = model.decoder.word_embed
        for param in self.decoder.word_embed.parameters():
            param.requires_grad = False
        self._flat = None                   # parameters changed identity: re-pack

    # ------------------------------------------------------------------ engine schedules
    def _encode(self, frames, regions, training, seed, sv):
        """CapGnnEncoder.forward (models/model.py:69-73).  The frame nodes of both streams are computed first (the motion
        stream's come out of the BiLSTM pre-encoder), then the object->frame graph of BOTH streams runs as one launch."""
        ops, enc = self.ops, self.encoder
        B, T, F = frames.shape
        A = enc.a_feature_size
        f2 = frames.view(B * T, F)
        # the object stream's visual_embed (independent of everything else) rides in the pre-encoder's first launch
        pre, extra = {}, []
        oe = enc.obj_encoder
        if oe.use_embed:
            pre['encoder.obj_encoder'] = torch.empty(B * T, oe.visual_embed.weight.shape[0], dtype=torch.float32, device=frames.device)
            extra.append((f2[:, :A], oe.visual_embed.weight, pre['encoder.obj_encoder'], oe.visual_embed.bias))
        mot_in = E.encvis_fwd(ops, enc.motion_pre_encoder, 'encoder.motion_pre_encoder', f2, B, T, sv, training, seed, extra=extra)
        E.tun_frames_multi(ops, [(enc.obj_encoder, 'encoder.obj_encoder', f2[:, :A]), (enc.motion_encoder, 'encoder.motion_encoder', mot_in)],
                           regions, sv, pre)
        # the region projections (the step's longest launch) go HERE, behind the frame path's matrix kernels, not first in the
        # step: directly behind the previous step's Adam -- 0.5 ms of pure memory traffic -- the same launch takes 11 % longer
        # (1 722 us against 1 555 back to back, tools/archive/sk_sequence_probe.py: the clock the chip holds, not the caches)
        ys = [None, None]
        if regions.shape[2] >= 5 and enc.obj_encoder.obj_embed.weight.shape == enc.motion_encoder.obj_embed.weight.shape:
            ys = E.region_projections(ops, [enc.obj_encoder, enc.motion_encoder], regions)
        E.tun_graph(ops, [(enc.obj_encoder, 'encoder.obj_encoder', ys[0]), (enc.motion_encoder, 'encoder.motion_encoder', ys[1])],
                    regions, sv, self.fused_o2v)
        obj, mot = E.tun_latent_multi(ops, [(enc.obj_encoder, 'encoder.obj_encoder', E.SITE_PSL_OBJ),
                                            (enc.motion_encoder, 'encoder.motion_encoder', E.SITE_PSL_MOT)], regions, sv, training, seed)
        return obj, mot

    def _engine_forward(self, frames, regions, captions, L, coins, training, seed, sv, dev_coins=None, outputs=True):
        ops = self.ops
        self._gemm_policy(False)
        if getattr(ops, 'colsum_defer', None) is not None:
            ops.colsum_defer = None         # (a backward that raised half-way must not leave the collector armed)
        frames = frames.contiguous().float()
        regions = regions.contiguous().float()
        obj, mot = self._encode(frames, regions, training, seed, sv)
        sv['frames'], sv['regions'] = frames, regions
        sv['dec_gsrc'] = [obj, mot]
        s = E.dec_fwd(ops, self.decoder, [obj, mot], sv, captions, L, coins, training, seed, dev_coins)
        if not outputs:            # fused trainer: the loss reads the time-major logits in place
            return None
        B = frames.shape[0]
        V = self.decoder.vocab_size
        logits = torch.empty(B, L, V, dtype=torch.float32, device=frames.device)
        ops.permute_tb(s['LOGITS'], logits)
        alpha = torch.empty(B, L, s['ALPHA'].shape[-1], dtype=torch.float32, device=frames.device)
        ops.permute_tb(s['ALPHA'], alpha)
        return logits, obj, mot, alpha

    def _engine_backward(self, sv, dlogits_tm, dobj, dmot, dalpha_tm, training, seed, on_bucket=None):
        ops, enc = self.ops, self.encoder
        self._gemm_policy(True)
        G = self._G
        ops.fill(self._gflat, 0.0)
        if self.merge_weight_grads and self._defer_ok:
            sv['tn_defer'] = []             # mid-size weight gradients of every module: launched together at the end
        collect = hasattr(ops, 'colsum_flush')
        if collect:
            ops.colsum_defer = []           # short column sums (LayerNorm / bias gradients): grouped launches (hip.py)
        frames, regions = sv['frames'], sv['regions']
        B, T, F = frames.shape
        H = self.decoder.visual_hidden_size
        dmems, dgfeat = E.dec_bwd(ops, self.decoder, sv, G, dlogits_tm, seed, training, dalpha_tm)

        def bucket(key):
            # a bucket's gradients must be complete when it is handed to the all-reduce: the weight gradients deferred so far
            # go out now, as one grouped launch per height (with one process nothing is reduced and they wait for the end)
            if on_bucket:
                if self._flush_at_buckets:
                    if sv.get('tn_defer'):
                        E.tn_grouped(ops, sv['tn_defer'])
                        sv['tn_defer'] = []
                    if collect:
                        ops.colsum_flush(keep_collecting=True)
                on_bucket(key)
        bucket('decoder')
        dob, dmo = dmems
        ops.mean_rows_bwd(dgfeat[:, :H], dob, accum=True)
        ops.mean_rows_bwd(dgfeat[:, H:], dmo, accum=True)
        if dobj is not None:
            ops.copy2d(dobj.reshape(-1, H), dob.view(-1, H), accum=True)
        if dmot is not None:
            ops.copy2d(dmot.reshape(-1, H), dmo.view(-1, H), accum=True)
        f2 = frames.view(B * T, F)
        # the obj_embed weight gradients of the two streams (the deepest products of the step) are issued together at the
        # end, so the two small TUN buckets (12 MB each) are reduced last; the 151 MB motion_pre_encoder bucket still
        # travels under the object stream's backward
        deep = []
        mot, obj = (enc.motion_encoder, 'encoder.motion_encoder'), (enc.obj_encoder, 'encoder.obj_encoder')
        # both streams' LatentPSL / obj_visual_norm backward first, so that the object->frame graph of BOTH streams runs its
        # backward as one launch per pass (2 x 64 clips: two object chunks per clip instead of four)
        E.tun_bwd_head_multi(ops, [(mot[0], mot[1], dmo), (obj[0], obj[1], dob)], regions, sv, G, training, seed)
        E.tun_graph_bwd(ops, [mot, obj], regions, sv, G)
        dmot_in = E.tun_bwd_tail(ops, mot[0], mot[1], regions, sv, G, defer_dw=deep)
        E.encvis_bwd(ops, enc.motion_pre_encoder, 'encoder.motion_pre_encoder', f2, B, T, sv, G, dmot_in, training, seed)
        bucket('encoder.motion_pre_encoder')
        E.tun_bwd_tail(ops, obj[0], obj[1], regions, sv, G, defer_dw=deep)
        if on_bucket and self._flush_at_buckets and len(deep) == 2:
            # several ranks: what trails the backward is exposed.  Everything of the two graph modules but their obj_embed weights
            # is complete here and travels under the deep weight-gradient products (1.8 ms); those run one stream at a time, so the
            # motion stream's 8 MB ride under the object stream's product and only obj_encoder.obj_embed.weight (8 MB of the 25)
            # is reduced after the last launch of the backward.  (One rank: the two products share ONE launch.)
            bucket(('encoder.motion_encoder:rest', 'encoder.obj_encoder:rest'))
            for item, (_, pfx) in zip(deep, (mot, obj)):
                E.gemm_tn_deep(ops, [item], frames)
                on_bucket(pfx + ':obj_embed.weight')
            if collect:
                ops.colsum_flush()
            return
        E.gemm_tn_deep(ops, deep, frames)
        if 'tn_defer' in sv:
            E.tn_grouped(ops, sv.pop('tn_defer'))
        if collect:
            ops.colsum_flush()
        if on_bucket:
            on_bucket(('encoder.motion_encoder', 'encoder.obj_encoder'))

    # ------------------------------------------------------------------ public forward
    def forward(self, visual_feats, region_feats, caption, max_words=None, teacher_forcing_ratio=1.0):
        self.flatten_parameters_()
        dec = self.decoder
        infer = caption is None
        L = dec.max_words if max_words is None else max_words
        if infer and dec.beam_size != 1:
            from .beam import beam_infer
            return beam_infer(self, visual_feats, region_feats)
        coins = self._draw_coins(L, infer, teacher_forcing_ratio)
        seed = self.next_seed()
        if infer:
            sv = {}
            with torch.no_grad():
                self._engine_forward(visual_feats, region_feats, None, L, coins, False, seed, sv)
            ids = sv['dec']['IDS'][1:].t().contiguous()
            return ids, sv['dec_gsrc'][0], sv['dec_gsrc'][1], []
        params = [p for _, p in self.named_parameters()]
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        if not needs_grad:
            with torch.no_grad():
                return self._engine_forward(visual_feats, region_feats, caption, L, coins, self.training, seed, {})
        return _ModelFn.apply(self, visual_feats, region_feats, caption, L, coins, seed, *params)


class CapBaseline1(_HipModel):
    """models/model.py:94-107: frames-only variant, EncoderVisual(baseline) + Decoder(baseline)."""
    unused_parameters = frozenset(['decoder.context_layernorm.weight', 'decoder.context_layernorm.bias'])

    def __init__(self, args, vocab):
        super().__init__()
        self.check_kernel_limits(args, args.max_frames, 'max_frames (baseline decoders attend over the frame nodes)')
        self.use_visual_gan = args.use_visual_gan
        self.encoder = EncoderVisual(args, baseline=True)
        self.decoder = Decoder(args, vocab, multi_modal=False, baseline=True)

    def update_beam_size(self, beam_size):
        self.decoder.update_beam_size(beam_size)

    def _engine_forward(self, frames, regions, captions, L, coins, training, seed, sv, dev_coins=None, outputs=True):
        ops = self.ops
        self._gemm_policy(False)
        frames = frames.contiguous().float()
        B, T, F = frames.shape
        H = self.decoder.visual_hidden_size
        enc = E.encvis_fwd(ops, self.encoder, 'encoder', frames.view(B * T, F), B, T, sv, training, seed).view(B, T, H)
        sv['frames'] = frames
        sv['dec_gsrc'] = [enc]
        s = E.dec_fwd(ops, self.decoder, [enc], sv, captions, L, coins, training, seed, dev_coins)
        V = self.decoder.vocab_size
        logits = torch.empty(B, L, V, dtype=torch.float32, device=frames.device)
        ops.permute_tb(s['LOGITS'], logits)
        return logits, enc, enc, torch.empty(0, device=frames.device)

    def _engine_backward(self, sv, dlogits_tm, dobj, dmot, dalpha_tm, training, seed, on_bucket=None):
        ops = self.ops
        self._gemm_policy(True)
        G = self._G
        ops.fill(self._gflat, 0.0)
        frames = sv['frames']
        B, T, F = frames.shape
        dmems, dgfeat = E.dec_bwd(ops, self.decoder, sv, G, dlogits_tm, seed, training, None)
        denc = dmems[0]
        ops.mean_rows_bwd(dgfeat, denc, accum=True)
        E.encvis_bwd(ops, self.encoder, 'encoder', frames.view(B * T, F), B, T, sv, G, denc.view(B * T, -1), training, seed)
        if on_bucket:
            on_bucket(('decoder', 'encoder'))

    def forward(self, visual_feats, region_feats, caption, max_words=None, teacher_forcing_ratio=1.0):
        self.flatten_parameters_()
        dec = self.decoder
        infer = caption is None
        L = dec.max_words if max_words is None else max_words
        if infer and dec.beam_size != 1:
            from .beam import beam_infer
            return beam_infer(self, visual_feats, region_feats)
        coins = self._draw_coins(L, infer, teacher_forcing_ratio)
        seed = self.next_seed()
        if infer:
            sv = {}
            with torch.no_grad():
                self._engine_forward(visual_feats, region_feats, None, L, coins, False, seed, sv)
            return sv['dec']['IDS'][1:].t().contiguous(), 0, 0, 0
        params = [p for _, p in self.named_parameters()]
        if not (torch.is_grad_enabled() and any(p.requires_grad for p in params)):
            with torch.no_grad():
                return self._engine_forward(visual_feats, region_feats, caption, L, coins, self.training, seed, {})[0], 0, 0, 0
        out = _ModelFn.apply(self, visual_feats, region_feats, caption, L, coins, seed, *params)
        return out[0], 0, 0, 0


class CapBaselineModel(_HipModel):
    """models/model.py:76-91: CapGnnEncoder(baseline=True) -- the TUN streams return their frame nodes instead of latent
    proposals -- and a baseline Decoder (one attention) on the motion stream only.  The reference also runs the object
    stream and drops its output; that work is skipped here (no output or gradient depends on it), and the 24 parameters
    the reference leaves without a gradient (object stream, LatentPSL of the motion stream, linear_baseline, ...) are
    reported as unused."""

    @property
    def unused_parameters(self):
        names = ['decoder.context_layernorm.weight', 'decoder.context_layernorm.bias', 'linear_baseline.weight',
                 'linear_baseline.bias']
        for n, _ in self.encoder.obj_encoder.named_parameters():
            names.append('encoder.obj_encoder.' + n)
        me = 'encoder.motion_encoder.'
        names += [me + 'att_l2l_norm.weight', me + 'att_l2l_norm.bias', me + 'v2l_layer.theta',
                  me + 'v2l_layer.out_norm.1.weight', me + 'v2l_layer.out_norm.1.bias']
        if not self.encoder.motion_encoder.has_obj:
            names += [me + 'obj_visual_norm.1.weight', me + 'obj_visual_norm.1.bias']
        return frozenset(names)

    def __init__(self, args, vocab):
        super().__init__()
        self.check_kernel_limits(args, args.max_frames, 'max_frames (baseline decoders attend over the frame nodes)')
        self.use_visual_gan = args.use_visual_gan
        self.encoder = CapGnnEncoder(args, baseline=True)
        self.linear_baseline = nn.Linear(args.visual_hidden_size * 2, args.visual_hidden_size)
        self.decoder = Decoder(args, vocab, multi_modal=False, baseline=True)

    def update_beam_size(self, beam_size):
        self.decoder.update_beam_size(beam_size)

    def _motion_nodes(self, frames, regions, training, seed, sv):
        ops, enc = self.ops, self.encoder
        B, T, F = frames.shape
        f2 = frames.view(B * T, F)
        mot_in = E.encvis_fwd(ops, enc.motion_pre_encoder, 'encoder.motion_pre_encoder', f2, B, T, sv, training, seed)
        mot = E.tun_fwd(ops, enc.motion_encoder, 'encoder.motion_encoder', mot_in, regions, sv, training, seed,
                        E.SITE_PSL_MOT, self.fused_o2v)
        return mot.view(B, T, -1)

    def _engine_forward(self, frames, regions, captions, L, coins, training, seed, sv, dev_coins=None, outputs=True):
        ops = self.ops
        self._gemm_policy(False)
        frames = frames.contiguous().float()
        regions = regions.contiguous().float()
        mot = self._motion_nodes(frames, regions, training, seed, sv)
        sv['frames'], sv['regions'] = frames, regions
        sv['dec_gsrc'] = [mot]
        s = E.dec_fwd(ops, self.decoder, [mot], sv, captions, L, coins, training, seed, dev_coins)
        B = frames.shape[0]
        logits = torch.empty(B, L, self.decoder.vocab_size, dtype=torch.float32, device=frames.device)
        ops.permute_tb(s['LOGITS'], logits)
        return logits, mot, mot, torch.empty(0, device=frames.device)

    def _engine_backward(self, sv, dlogits_tm, dobj, dmot, dalpha_tm, training, seed, on_bucket=None):
        ops, enc = self.ops, self.encoder
        self._gemm_policy(True)
        G = self._G
        ops.fill(self._gflat, 0.0)
        frames, regions = sv['frames'], sv['regions']
        B, T, F = frames.shape
        dmems, dgfeat = E.dec_bwd(ops, self.decoder, sv, G, dlogits_tm, seed, training, None)
        dmo = dmems[0]
        ops.mean_rows_bwd(dgfeat, dmo, accum=True)
        dmot_in = E.tun_bwd(ops, enc.motion_encoder, 'encoder.motion_encoder', regions, sv, G, dmo, training, seed)
        E.encvis_bwd(ops, enc.motion_pre_encoder, 'encoder.motion_pre_encoder', frames.view(B * T, F), B, T, sv, G, dmot_in,
                     training, seed)
        if on_bucket:
            on_bucket(('decoder', 'linear_baseline', 'encoder.obj_encoder', 'encoder.motion_pre_encoder', 'encoder.motion_encoder'))

    forward = CapBaseline1.forward


def _h2d(values, dtype, device):
    from .hip import host_to_device
    return host_to_device(values, dtype, device)


def _copy_h2d(dst, values):
    from .hip import copy_to_device
    copy_to_device(dst, values)


def _capture_stream(dev):
    """the one side stream per device all hipGraph captures of this process run on (hip.HipOps.capture_stream)"""
    from .hip import HipOps
    return HipOps.capture_stream(dev)


class GreedyGraph(object):
    """hipGraph-captured greedy inference (BASELINE configs[4]: 'hipGraph-captured decode step'): encoder + the 26
    decode steps (argmax and embedding gather stay on device) are captured once for a batch shape and replayed; the
    ids are identical to the eager `model(frames, regions, None)` path with beam_size 1."""

    def __init__(self, model, frames, regions):
        self.model = model
        model.flatten_parameters_()
        dev = frames.device
        self.frames, self.regions = frames.clone(), regions.clone()
        L = model.decoder.max_words
        side = _capture_stream(dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            model._engine_forward(self.frames, self.regions, None, L, [False] * L, False, 0, {})     # warm-up
            side.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            self.graph.capture_begin(capture_error_mode='thread_local')
            sv = {}
            model._engine_forward(self.frames, self.regions, None, L, [False] * L, False, 0, sv)
            self.ids = sv['dec']['IDS'][1:].t().contiguous()
            self.graph.capture_end()
        torch.cuda.current_stream().wait_stream(side)

    @torch.no_grad()
    def __call__(self, frames, regions):
        if frames.data_ptr() != self.frames.data_ptr():
            self.frames.copy_(frames, non_blocking=True)
        if regions.data_ptr() != self.regions.data_ptr():
            self.regions.copy_(regions, non_blocking=True)
        self.graph.replay()
        return self.ids


class BeamGraph(object):
    """hipGraph-captured beam search (BASELINE configs[4]): encoder + all max_words beam steps (decode step, `beam_select`,
    state reorder) captured once for a batch shape; a replay has no host synchronisation, the early stop of the reference is
    applied afterwards (`beam.beam_finish`).  Ids are identical to `model(frames, regions, None)` with the same beam size."""

    def __init__(self, model, frames, regions):
        from .beam import beam_device
        self.model = model
        model.flatten_parameters_()
        dev = frames.device
        self.frames, self.regions = frames.clone(), regions.clone()
        side = _capture_stream(dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            beam_device(model, self.frames, self.regions, early_exit=False)                       # warm-up
            side.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            self.graph.capture_begin(capture_error_mode='thread_local')
            self.state = beam_device(model, self.frames, self.regions, early_exit=False)
            self.graph.capture_end()
        torch.cuda.current_stream().wait_stream(side)

    @torch.no_grad()
    def __call__(self, frames, regions):
        from .beam import beam_finish
        if frames.data_ptr() != self.frames.data_ptr():
            self.frames.copy_(frames, non_blocking=True)
        if regions.data_ptr() != self.regions.data_ptr():
            self.regions.copy_(regions, non_blocking=True)
        self.graph.replay()
        return beam_finish(self.model, *self.state)


# ================================================================================================ fast training path
def ss_epsilon(epoch, ss_factor=20):
    """scheduled-sampling probability, run_gun.py:136"""
    return max(0.6, ss_factor / (ss_factor + math.exp(epoch / ss_factor)))


def multistep_lr(epoch, base_lr=1.6e-4, milestones=(4, 7), gamma=0.5):
    """learning rate of `MultiStepLR(optimizer, milestones=[4, 7], gamma=0.5)` (run_gun.py:94-95) at the start of `epoch`;
    assign it to `Trainer.lr` (the replayed graphs read the rate from device memory every step)."""
    return base_lr * gamma ** sum(1 for m in milestones if epoch >= m)


class Trainer(object):
    """The reference's per-iteration use of the model (run_gun.py:153-160,181-198,233-234) as one fused schedule:
    forward -> ragged CrossEntropy (mean over sum(cap_lens) rows) -> backward -> [RCCL all-reduce] -> Adam.

    Data parallel: one process per GPU; every rank runs its own shard, gradients are summed with
    torch.distributed all_reduce (RCCL over xGMI) bucket by bucket as the backward finishes each module group and
    divided by world size inside the Adam kernel (DDP mean-of-means semantics, run_gun.py:63-64).

    use_graphs: capture the step into hipGraphs (one per backward bucket, so the collectives stay between graph
    replays and still overlap the rest of the backward).  The step is ~1200 kernel launches, i.e. launch-bound from
    Python; replaying removes the host from the loop.  Everything that changes between steps is read from device
    memory: inputs (static buffers), the dropout seed, the scheduled-sampling coins, the Adam bias corrections."""

    def __init__(self, model, lr=1.6e-4, betas=(0.5, 0.9), eps=1e-8, process_group=None, world_size=1, use_graphs=False,
                 device_coins=None, graph_fallback=False, comm='auto', check_every=100, rehearse_ranks=0):
        """comm: how the gradient buckets are summed over ranks.
          'rccl'  -- librccl through the C ABI (dlsg_allreduce_bucket) on a side stream forked by an event: the collectives
                     are part of the captured step, so one iteration is ONE hipGraph replay and a bucket's all-reduce runs
                     under the backward that follows it;
          'torch' -- torch.distributed.all_reduce(async_op=True) issued by the host between hipGraph segments (any backend:
                     this is what the CPU / gloo tests run);
          'auto'  -- 'rccl' when the model lives on a GPU, else 'torch'.
        rehearse_ranks: N > 1 with world_size == 1 runs, on ONE GPU, exactly the step a rank of an N-rank job runs -- the
          multi-rank kernel choices of `_use_multi_rank_schedule`, weight gradients flushed at every bucket, every bucket handed
          to the communicator (world 1: the all-reduce is the identity, Adam's 1 / world stays 1) -- so that this schedule can be
          checked against the oracle and timed where only one device exists (tests/test_gpu_bench_parity.py, bench.py's
          `dp_schedule_world1`).  `rehearse_cotenant` adds the chip-sharing a real all-reduce brings."""
        assert comm in ('auto', 'rccl', 'torch'), comm
        self.comm = comm
        # every `check_every` steps the persistent kernels' time-out word is read back (one host synchronisation): a launch that
        # was not co-resident (GPU shared with another process) raises here; meanwhile dlsg_adam's guard kept the weights intact
        self.check_every = check_every
        self._rccl = None
        self._comm_stream = None
        self._comm_pending = False
        self.model = model
        self.lr, self.betas, self.eps = lr, betas, eps
        self.t = 0
        self.world_size = world_size
        self.pg = process_group
        self.use_graphs = use_graphs
        self.device_coins = use_graphs if device_coins is None else device_coins
        self.graph_fallback = graph_fallback   # True: a failed capture downgrades to eager launches (with a warning)
        self.force_graph_cuts = False   # test hook: segment the capture at bucket boundaries even on one GPU
        self.force_collectives = False  # test hook: issue the all-reduces even with one rank (RCCL path on a 1-GPU box)
        # rehearsal hook (one GPU): None, or dict(workgroups=32, passes=n) -- behind every bucket's all-reduce the side stream also
        # runs dlsg_comm_rehearsal over the bucket: `workgroups` x 256 threads streaming it `passes` times, i.e. the CUs and the
        # HBM share a ring all-reduce over xGMI would hold while the backward goes on (values unchanged)
        self.rehearse_cotenant = None
        self.rehearse_ranks = int(rehearse_ranks)
        self._works = []
        self._graphs = None
        self._contact_checked = False
        self._hook_mode, self._hook_sv = False, None
        self.m = self.v = None
        if world_size > 1 or self.rehearse_ranks > 1:
            assert world_size == 1 or not self.rehearse_ranks, 'rehearse_ranks is for one-rank runs'
            self._use_multi_rank_schedule()
            self.force_collectives = self.force_collectives or self.rehearse_ranks > 1
        self._bind()

    # CUs a bucket's all-reduce holds while the backward runs beside it: RCCL launches one 256-thread workgroup per channel (up
    # to 32 on this part).  What a stream-K launch would be budgeted to leave free (`model.sk_backward_cu_budget = cus - COMM_CUS`).
    COMM_CUS = 32

    def _use_multi_rank_schedule(self):
        """The kernel choices of a rank that shares its GPU with gradient all-reduces (set once, at construction, for
        world_size > 1 and for one-GPU rehearsals of it; run_gun.py:63-64 leaves all of this to DDP + NCCL).
        * The persistent BiLSTM kernels need all of their workgroups (one per CU at H = 1024, 152 KB of LDS each) resident
          together.  The encoder's backward runs while the decoder bucket's all-reduce is in flight: a CU that hosts an RCCL
          workgroup has no room for a 152-KB one, so the launch would sit half-resident, its workgroups polling for partners that
          cannot start, until the collective ends -- no deadlock (RCCL does not depend on it), but the CUs it holds and the overlap
          window are lost.  The backward through time therefore runs step by step (0.60 against 0.50 ms); the forward keeps the
          persistent launch (no collective is in flight there: the previous step's all-reduces are joined before its Adam).
        * The stream-K GEMM (csrc/gemm_sk.hip) is one workgroup per CU with the whole register file of its SIMDs, so beside a
          collective its last workgroups start only when CUs come free.  Round 5 put the backward on the tiled kernels for that
          reason, unmeasured.  Round 6 measured it (bench.py `dp_schedule_world1`, DESIGN.md section 6: this schedule on one GPU
          with a co-tenant of RCCL's launch shape -- 32 x 256 threads streaming every bucket for the ~1.1 ms per 180 MB an 8-rank
          ring takes -- on the side stream): stream-K on every CU 14.79 ms per step, tiled backward 15.07, stream-K on a budget
          of 224 workgroups (the 32 CUs left free) 15.33; batch 128: 24.17 / 25.39 / 25.10; without the co-tenant 12.99 / 13.38 /
          13.49 against 12.93 for the one-rank schedule.  The kernel never waits for a workgroup that has not started (shares
          are given away, csrc/gemm_sk.hip), so a launch that is not co-resident is slower by what the co-tenant holds, not by a
          round.  Hence: the backward KEEPS the stream-K launches, on every CU (`model.sk_backward_cu_budget = 0`); the budget
          and `model.stream_k_in_backward = False` remain as switches."""
        model = self.model
        if hasattr(model.ops, 'persistent_bilstm_bwd'):
            model.ops.persistent_bilstm_bwd = False
        model.stream_k_in_backward = True
        model.sk_backward_cu_budget = 0

    def _bind(self):
        """(Re)attach to the model's arenas: Adam moments, bucket ranges, trainable ranges.  Runs again whenever the model
        re-packed its parameters (load_encoder grafts modules) or a parameter's requires_grad changed."""
        model = self.model
        model.flatten_parameters_()
        old_m, old_v = self.m, self.v
        self.m = torch.zeros_like(model._flat)
        self.v = torch.zeros_like(model._flat)
        if old_m is not None and old_m.shape == self.m.shape and old_m.device == self.m.device:
            self.m.copy_(old_m); self.v.copy_(old_v)      # same layout (only the frozen set changed): keep the moments
        self._arena = model._flat
        self._graphs = None
        # contiguous arena range of each backward bucket (named_parameters order == arena order)
        self._ranges = {}
        frozen = []
        for name, p in model.named_parameters():
            key = name.split('.')[0] if not name.startswith('encoder.') or not hasattr(model.encoder, 'obj_encoder') \
                else '.'.join(name.split('.')[:2])
            o = model._offsets[name]
            end = o + (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
            lo, hi = self._ranges.get(key, (o, end))
            self._ranges[key] = (min(lo, o), max(hi, end))
            if not p.requires_grad:
                frozen.append((o, end))
            if name.endswith('_encoder.obj_embed.weight') and key != name:
                # the deepest weight gradient of a graph module is its last: its range and the rest of the module are buckets of
                # their own (the weight is the module's first parameter: the rest is one contiguous range behind it)
                self._ranges[key + ':obj_embed.weight'] = (o, end)
        for key in [k for k in self._ranges if k.endswith(':obj_embed.weight')]:
            mod = key[:-len(':obj_embed.weight')]
            (wlo, whi), (mlo, mhi) = self._ranges[key], self._ranges[mod]
            if wlo == mlo:
                self._ranges[mod + ':rest'] = (whi, mhi)
            else:
                del self._ranges[key]
        self._frozen = tuple(frozen)
        # maximal runs of trainable parameters: Adam and the all-reduces cover exactly these (torch.optim skips parameters
        # without a gradient; DDP does not reduce them)
        self._train_ranges = self._minus_frozen(0, model._flat.numel())

    def _minus_frozen(self, lo, hi):
        out, cur = [], lo
        for a, b in self._frozen:
            if b <= lo or a >= hi:
                continue
            if a > cur:
                out.append((cur, a))
            cur = max(cur, b)
        if cur < hi:
            out.append((cur, hi))
        return out

    def _check_binding(self):
        model = self.model
        model.flatten_parameters_()
        frozen = tuple((model._offsets[n], model._offsets[n] + (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN)
                       for n, p in model.named_parameters() if not p.requires_grad)
        if self._arena is not model._flat or frozen != self._frozen:
            self._bind()

    def _adam(self, step, hyper=None, ranges=None):
        model, ops = self.model, self.model.ops
        for lo, hi in (self._train_ranges if ranges is None else ranges):
            ops.adam(model._flat[lo:hi], model._gflat[lo:hi], self.m[lo:hi], self.v[lo:hi], self.lr, self.betas[0],
                     self.betas[1], self.eps, step, 1.0 / self.world_size, hyper=hyper)

    # ------------------------------------------------------------------ checkpoint compatibility (run_gun.py:302-310)
    def optimizer_state_dict(self):
        """The Adam state in the layout of `torch.optim.Adam(model.parameters(), ...).state_dict()` -- what the reference
        stores as `optimizer_state_dict` in its checkpoints.  Parameters that never receive a gradient have no entry,
        as in torch (Adam creates state lazily for parameters with a gradient)."""
        model = self.model
        state = {}
        names = [n for n, _ in model.named_parameters()]
        for i, (name, p) in enumerate(model.named_parameters()):
            if name in model.unused_parameters or not p.requires_grad or self.t == 0:
                continue
            o = model._offsets[name]
            state[i] = {'step': torch.tensor(float(self.t)),
                        'exp_avg': self.m[o:o + p.numel()].view(p.shape).clone(),
                        'exp_avg_sq': self.v[o:o + p.numel()].view(p.shape).clone()}
        group = {'lr': self.lr, 'betas': tuple(self.betas), 'eps': self.eps, 'weight_decay': 0, 'amsgrad': False,
                 'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
                 'decoupled_weight_decay': False, 'params': list(range(len(names)))}
        return {'state': state, 'param_groups': [group]}

    def load_optimizer_state_dict(self, sd):
        """Resume from a reference checkpoint's `optimizer_state_dict` (torch.optim.Adam layout)."""
        model = self.model
        model.flatten_parameters_()
        self.m.zero_()
        self.v.zero_()
        steps = set()
        for i, (name, p) in enumerate(model.named_parameters()):
            st = sd['state'].get(i, sd['state'].get(str(i)))
            if st is None:
                continue
            o = model._offsets[name]
            self.m[o:o + p.numel()].view(p.shape).copy_(st['exp_avg'])
            self.v[o:o + p.numel()].view(p.shape).copy_(st['exp_avg_sq'])
            steps.add(int(float(st['step'])))
        if len(steps) > 1:
            raise ValueError('per-parameter step counts differ (%s): not a state this trainer can resume' % sorted(steps))
        self.t = steps.pop() if steps else 0
        g = sd['param_groups'][0]
        self.lr, self.betas, self.eps = g['lr'], tuple(g['betas']), g['eps']
        self._graphs = None          # captured graphs bake nothing of this state, but recapture keeps the invariants simple

    # ------------------------------------------------------------------ collectives
    def _comm_mode(self):
        """'none' | 'rccl' | 'torch' for this trainer's next step"""
        if self.world_size <= 1 and not self.force_collectives:
            return 'none'
        if self.comm == 'torch':
            return 'torch'
        on_gpu = self.model._flat is not None and self.model._flat.is_cuda
        if self.comm == 'rccl' and not on_gpu:
            raise RuntimeError("comm='rccl' needs the model on a GPU")
        return 'rccl' if on_gpu else 'torch'

    def _rccl_comm(self):
        if self._rccl is None:
            from .comm import RcclComm
            rank = 0
            if self.world_size > 1:
                import torch.distributed as dist
                rank = dist.get_rank(self.pg)
            self._rccl = RcclComm(self.world_size, rank, self.pg, lib=getattr(self.model.ops, 'lib', None))
            self._side_stream()
            self._first_contact(rank)
        return self._rccl

    def _first_contact(self, rank):
        """One eager all-reduce of a rank-id pattern through the new communicator before anything is captured: every element must
        come back as 0 + 1 + ... + (N - 1) on every rank.  A communicator that is wired wrongly (a rank on the wrong device, two
        jobs sharing an id, a transport that drops data) is found here, with the list of ranks that saw a wrong sum, instead
        of as diverging replicas or a hang inside a replayed graph (run_gun.py:63-64 relies on DDP's own constructor checks)."""
        if self.world_size <= 1:
            return
        import torch.distributed as dist
        dev = self.model._flat.device
        side = self._side_stream()
        t = torch.full((4096,), float(rank), dtype=torch.float32, device=dev)
        side.wait_stream(torch.cuda.current_stream())
        self._rccl.allreduce([t], side)
        side.synchronize()
        want = self.world_size * (self.world_size - 1) / 2.0
        ok = bool((t == want).all().item())
        every = [None] * self.world_size
        dist.all_gather_object(every, (rank, ok, float(t[0].item())), group=self.pg)
        bad = [(r, v) for r, o, v in every if not o]
        if bad:
            raise RuntimeError('RCCL first-contact check failed: the all-reduced rank pattern should be %g everywhere; wrong on ranks %s'
                               % (want, ', '.join('%d (got %g)' % rv for rv in bad)))

    def _side_stream(self):
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=self.model._flat.device)
        return self._comm_stream

    def _allreduce(self, key):
        """key: a bucket name or a tuple of bucket names whose gradients are complete: hand them to the reduction.  (Adam per
        bucket on the side stream right behind its all-reduce was measured and removed: 14.87 ms per step against 14.47 -- the
        update's 1.4 GB stream evicts the Infinity-Cache-resident operands of the backward that is still running.)"""
        mode = self._comm_mode()
        ranges = [r for k in (key if isinstance(key, tuple) else (key,)) for r in self._minus_frozen(*self._ranges[k])]
        views = [self.model._gflat[lo:hi] for lo, hi in ranges]
        if mode == 'torch':
            import torch.distributed as dist
            for v in views:
                self._works.append(dist.all_reduce(v, group=self.pg, async_op=True))
            return
        if mode == 'none':
            return
        comm = self._rccl_comm()
        side = self._side_stream()
        # fork: the side stream waits for everything enqueued so far (the bucket's gradients), the main stream goes on
        # with the rest of the backward; under stream capture both become edges of the step's graph
        ev = torch.cuda.Event()
        ev.record()
        side.wait_event(ev)
        comm.allreduce(views, side)
        if self.rehearse_cotenant and self.world_size == 1:
            co = self.rehearse_cotenant
            with torch.cuda.stream(side):
                for v in views:
                    self.model.ops.comm_rehearsal(v, int(co.get('workgroups', self.COMM_CUS)), int(co['passes']))
        self._comm_pending = True

    def _reduce_guard(self):
        """The persistent kernels' time-out word (BiLSTM, critic LSTM), max over the ranks, BEHIND the last launch of the backward
        and in front of Adam: dlsg_adam's guard then skips the update on every rank or on none -- a rank skipping alone would leave
        the replicas diverged, its invalid gradients already summed into everybody's.  (Until round 5 the word rode with the FIRST
        bucket: a time-out in the encoder's backward, behind that bucket, was not agreed.)  Four bytes behind the step's last,
        already exposed bucket on the same side stream."""
        mode = self._comm_mode()
        if mode == 'none':
            return
        # (the word is created here if no persistent kernel has run yet: whether a rank takes part in this collective must not
        #  depend on what its kernels happened to do)
        mk = getattr(self.model.ops, 'guard_word', None)
        if mk is None:
            return
        word = mk(self.model._flat.device)
        if mode == 'torch':
            import torch.distributed as dist
            self._works.append(dist.all_reduce(word, op=dist.ReduceOp.MAX, group=self.pg, async_op=True))
            return
        comm = self._rccl_comm()
        side = self._side_stream()
        ev = torch.cuda.Event()
        ev.record()
        side.wait_event(ev)
        comm.allreduce_max_word(word, side)
        self._comm_pending = True

    def _join_comm(self):
        """the main stream waits for the side stream's collectives (before Adam; before a capture segment ends)"""
        if self._comm_pending:
            ev = torch.cuda.Event()
            ev.record(self._comm_stream)
            torch.cuda.current_stream().wait_event(ev)
            self._comm_pending = False

    def collectives_info(self):
        mode = self._comm_mode()
        info = {'mode': {'none': 'none (one rank)', 'rccl': 'RCCL through the C ABI (dlsg_allreduce_bucket), side stream',
                         'torch': 'torch.distributed all_reduce(async_op=True) issued by the host'}[mode],
                'where': None if mode == 'none' else ('inside the step\'s hipGraph' if (mode == 'rccl' and self.use_graphs and
                                                      self._graphs is not None and len(self._graphs) == 1)
                                                     else ('between hipGraph segments' if self.use_graphs and self._graphs is not None
                                                           else 'between eager launches')),
                'graph_replays_per_step': len(self._graphs) if (self.use_graphs and self._graphs is not None) else 0,
                'buckets_MB': {k: round(4e-6 * sum(hi - lo for lo, hi in self._minus_frozen(*r)), 1) for k, r in self._ranges.items()}}
        if self._rccl is not None:
            info['rccl_version'] = self._rccl.rccl_version
        return info

    def check(self):
        """Raise if a persistent kernel's inter-workgroup wait timed out since the last call (ops.check_persistent).  A host
        synchronisation: call it where the loss is read back (per logging interval / epoch), not per step."""
        ops = self.model.ops
        chk = getattr(ops, 'check_persistent', None)
        if chk is None:
            return
        w = ops.persist_word_or_none()
        code = int(w.item()) if w is not None else 0
        if self._rccl is not None:
            ae = self._rccl.async_error()
            if ae:
                import sys
                sys.stderr.write('dlsg_amd.Trainer: RCCL reports asynchronous error %d on this rank (ncclCommGetAsyncError)\n' % ae)
                code = code or 100 + ae
        if self.world_size > 1:
            # every rank learns the worst code before any of them raises: one rank leaving alone would strand its peers in the
            # next step's in-graph all-reduce (a private communicator has no time-out)
            from .comm import _agree_min
            code = _agree_min(code, self.pg, negate=True)
        if code:
            self._graphs = None            # captured on the persistent schedule: the next step captures again
            if code >= 100:
                raise RuntimeError('RCCL asynchronous error %d on at least one rank (ncclCommGetAsyncError): the gradient exchange '
                                   'of the steps since the last check is not to be trusted' % (code - 100))
            if w is not None and int(w.item()) == 0:
                w.fill_(code)              # (another rank's time-out: the same exception here)
            chk(code=code)

    def close(self):
        """destroy the RCCL communicator (collective: every rank calls it)"""
        self.check()
        if self._rccl is not None:
            torch.cuda.synchronize()
            self._rccl.close()
            self._rccl = None

    # ------------------------------------------------------------------ one step, as a schedule
    def _schedule(self, frames, regions, captions, cap_lens, coins, seed, dev_coins, on_bucket, extra_dlogits=None):
        model, ops = self.model, self.model.ops
        L = captions.shape[1]
        sv = {}
        training = model.training
        # a bucket handed to a reduction (or closing a graph segment) must be complete: deferred weight gradients go out there
        model._flush_at_buckets = self._comm_mode() != 'none' or self.force_graph_cuts
        model._engine_forward(frames, regions, captions, L, coins, training, seed, sv, dev_coins, outputs=False)
        s = sv['dec']
        Bn = captions.shape[0]
        dl = torch.empty_like(s['LOGITS'])
        row_loss = torch.empty(L * Bn, dtype=torch.float32, device=dl.device)
        loss = torch.empty(1, dtype=torch.float32, device=dl.device)
        ops.ce_ragged(s['LOGITS'], captions, cap_lens, dl, row_loss, loss, time_major=True)
        if extra_dlogits is not None:
            # another loss on the logits (the GAN term of run_gun.py:218-231): its gradient joins the CrossEntropy's
            sv['loss_dev'] = loss
            g = extra_dlogits(s['LOGITS'], sv)
            V = dl.shape[-1]
            ops.copy2d(g.contiguous().view(-1, V), dl.view(-1, V), accum=True)
        model._engine_backward(sv, dl, None, None, None, training, seed, on_bucket=on_bucket)
        return loss

    def _hyper(self):
        b1, b2 = self.betas
        return [self.lr / (1.0 - b1 ** self.t), math.sqrt(1.0 - b2 ** self.t)]

    @torch.no_grad()
    def step(self, frames, regions, captions, cap_lens, tf_ratio, max_len=26, extra_dlogits=None):
        """One optimisation step.  Returns the (device) scalar loss of this rank's shard.
        extra_dlogits: optional callable (logits (L,B,V) time-major, saved-state dict) -> (L,B,V) gradient of an additional
        loss on the logits, added to the CrossEntropy's before the backward.  With use_graphs the captured step is cut at
        that point: forward + CrossEntropy replay, the callable runs (on the graphs' static logits / saved state), its result
        is copied into a static buffer that the replayed backward adds to the CrossEntropy gradient."""
        model, ops = self.model, self.model.ops
        self._check_binding()
        captions = captions[:, :max_len].contiguous()
        if not (torch.is_tensor(cap_lens) and cap_lens.device == captions.device and cap_lens.dtype == torch.int64):
            cap_lens = _h2d(cap_lens, torch.int64, captions.device)
        L = captions.shape[1]
        coins = model._draw_coins(L, False, tf_ratio)
        seed = model.next_seed()
        self.t += 1
        hook = extra_dlogits is not None
        if self.use_graphs and (self._graphs is None or self._hook_mode == hook):
            loss = self._step_graphs(frames, regions, captions, cap_lens, coins, seed, extra_dlogits)
        else:
            # (a trainer's graphs are captured for one of the two forms of the step; the other form runs kernel by kernel)
            loss = self._eager_step(frames, regions, captions, cap_lens, coins, seed, counted=True, extra_dlogits=extra_dlogits)
        if self.check_every and self.t % self.check_every == 0:
            self.check()
        return loss

    def _eager_step(self, frames, regions, captions, cap_lens, coins, seed, counted=False, extra_dlogits=None):
        model, ops = self.model, self.model.ops
        if not counted:
            self.t += 1
        dev_coins = None
        if self.device_coins:
            dev_coins = _h2d([int(c) for c in coins], torch.int32, captions.device)
        self._works = []
        loss = self._schedule(frames, regions, captions, cap_lens, coins, seed, dev_coins, self._allreduce, extra_dlogits)
        self._reduce_guard()
        for w in self._works:
            w.wait()
        self._join_comm()
        self._adam(self.t)
        return loss

    # ------------------------------------------------------------------ hipGraph path
    def _capture(self, frames, regions, captions, cap_lens, hook=False):
        dev = frames.device
        L = captions.shape[1]
        self._hook_mode, self._hook_sv = hook, None
        st = self._static = dict(frames=frames.clone(), regions=regions.clone(), captions=captions.clone(),
                                 lens=cap_lens.clone())
        # what changes every step besides the batch -- the scheduled-sampling coins, the dropout seed, Adam's bias corrections --
        # lives in ONE device buffer (int32 words: coins | seed (int64) | hyper (2 x float32)), so that a step sends it as one copy
        Lp = (L + 1) // 2 * 2
        st['scalars'] = torch.zeros(Lp + 4, dtype=torch.int32, device=dev)
        st['coins'] = st['scalars'][:L]
        st['coins'].fill_(1)
        st['seed'] = st['scalars'][Lp:Lp + 2].view(torch.int64)
        st['hyper'] = st['scalars'][Lp + 2:Lp + 4].view(torch.float32)
        side = _capture_stream(dev)
        side.wait_stream(torch.cuda.current_stream())
        graphs = []
        mode = self._comm_mode()
        # host-issued collectives need the capture cut at every bucket; RCCL's own launches are captured with the step
        cuts = mode == 'torch' or self.force_graph_cuts
        if mode == 'rccl':
            self._rccl_comm()              # communicator setup (a collective itself) outside any capture
        with torch.cuda.stream(side):
            # eager warm-up on the capture stream (allocator warm, one-time kernel attribute calls)
            # (with RCCL the warm-up also runs the collectives once: channel buffers are set up before the capture;
            #  nothing is updated: Adam is not part of the schedule)
            self._schedule(st['frames'], st['regions'], st['captions'], st['lens'], None, st['seed'], st['coins'],
                           self._allreduce if mode == 'rccl' else None)
            if mode == 'rccl':
                self._reduce_guard()
            self._join_comm()
            side.synchronize()
            if self._comm_stream is not None:
                self._comm_stream.synchronize()
            pool = torch.cuda.graph_pool_handle()
            # thread-local capture mode: calls made by other threads (e.g. the RCCL watchdog) cannot invalidate the capture
            cur = [torch.cuda.CUDAGraph()]
            cur[0].capture_begin(pool=pool, capture_error_mode='thread_local')
            try:
                def hard_cut(key, between=None):
                    cur[0].capture_end()
                    graphs.append((cur[0], key))
                    if between is not None:
                        between()
                    cur[0] = torch.cuda.CUDAGraph()
                    cur[0].capture_begin(pool=pool, capture_error_mode='thread_local')

                def cut(key):
                    if mode != 'torch':
                        self._allreduce(key)               # captured: fork to the side stream (collective and / or Adam)
                        if cuts:
                            self._join_comm()              # a capture segment must end with its forks joined
                    if cuts:
                        hard_cut(key if mode == 'torch' else None)

                def placeholder(logits, sv):
                    # the graphs end here and resume after the caller's term: what it will read (logits, saved state, the
                    # loss) lives in the graphs' pool, what it returns is copied into st['extra'] before the next replay
                    # (allocated BETWEEN the two captures: inside one, its zero fill would be replayed over the copy)
                    def alloc():
                        st['extra'] = torch.zeros_like(logits)
                    hard_cut('hook', alloc)
                    self._hook_sv = (logits, sv)
                    return st['extra']

                loss = self._schedule(st['frames'], st['regions'], st['captions'], st['lens'], None, st['seed'], st['coins'], cut,
                                      placeholder if hook else None)
                if mode != 'torch':
                    # no host-issued collective between backward and update: Adam is part of the graph, behind the join of the
                    # side stream's collectives; with host-issued collectives it follows their waits
                    self._reduce_guard()
                    self._join_comm()
                    self._adam(1, hyper=st['hyper'])
                cur[0].capture_end()
                graphs.append((cur[0], None))
            except BaseException:
                # leave no stream in capture mode behind (a later synchronize would raise on top of the real error) and
                # drop the partial graphs
                try:
                    cur[0].capture_end()
                except Exception:
                    pass
                graphs.clear()
                self._graphs = None
                self._comm_pending = False
                raise
        torch.cuda.current_stream().wait_stream(side)
        self._graphs, self._loss = graphs, loss
        self._adam_in_graph = mode != 'torch'

    def _capture_agreed(self, frames, regions, captions, cap_lens, hook):
        """`_capture`, and with several ranks the agreement on its outcome: returns None when EVERY rank captured, else the error
        (this rank's own, or a stand-in when only another rank failed) after dropping this rank's graphs -- so that all ranks take
        the same fallback together"""
        err = None
        try:
            self._capture(frames, regions, captions, cap_lens, hook)
        except Exception as e:            # (any failure votes: an AssertionError on one rank must not leave the others in the vote)
            err = e
        if self.world_size > 1:
            from .comm import _agree_min
            if _agree_min(0 if err is not None else 1, self.pg) == 0 and err is None:
                err = RuntimeError('hipGraph capture failed on another rank')
                self._graphs = None
        return err

    @torch.no_grad()
    def forward_only(self, frames, regions, captions, tf_ratio, max_len=26, time_major=False):
        """The no-grad generator forward of the GAN iteration (run_gun.py:167) from the captured step's FIRST graph (forward +
        CrossEntropy): same coin / dropout-seed draws as `model(frames, regions, captions, max_len, tf_ratio)`, returns
        (logits (B,L,V), obj, mot, alpha (B,L,2P)) as views of the graphs' static buffers -- valid until the next replay --
        or None when this trainer has no cut graphs for that batch shape (the caller then calls the model)."""
        if not (self.use_graphs and self._graphs is not None and self._hook_mode and self._hook_sv is not None
                and self.model.training):
            return None
        st = self._static
        captions = captions[:, :max_len].contiguous()
        if (frames.shape, regions.shape, captions.shape) != (st['frames'].shape, st['regions'].shape, st['captions'].shape):
            return None
        self._check_binding()
        model = self.model
        coins = model._draw_coins(captions.shape[1], False, tf_ratio)
        seed = model.next_seed()
        for k, src in (('frames', frames), ('regions', regions), ('captions', captions)):
            if src.data_ptr() != st[k].data_ptr():
                st[k].copy_(src, non_blocking=True)
        self._send_scalars(coins, seed, None)
        self._graphs[0][0].replay()
        logits_tm, sv = self._hook_sv
        # time_major: the logits as the decoder wrote them, (L,B,V) -- what the critic's schedule reads (gan.GanTrainer)
        return (logits_tm if time_major else logits_tm.transpose(0, 1)), sv['dec_gsrc'][0], sv['dec_gsrc'][1], \
            sv['dec']['ALPHA'].transpose(0, 1)

    def _send_scalars(self, coins, seed, hyper):
        """coins, dropout seed and (optionally) Adam's bias corrections of this step into the graphs' static words: one host buffer,
        one asynchronous copy (hyper None: the words keep their value)"""
        st = self._static
        n = st['scalars'].numel()
        L = st['coins'].numel()
        host = torch.empty(n, dtype=torch.int32)
        host[:L] = torch.tensor([int(c) for c in coins], dtype=torch.int32)
        host[L:n - 4] = 0
        host[n - 4:n - 2].view(torch.int64)[0] = seed
        if hyper is None:
            _copy_h2d(st['scalars'][:n - 2], host[:n - 2])
            return
        host[n - 2:].view(torch.float32).copy_(torch.tensor(hyper, dtype=torch.float32))
        _copy_h2d(st['scalars'], host)

    def static_inputs(self):
        """The captured graphs read their batch from these device buffers: (frames, regions, captions, cap_lens), or None
        before the first replayed step.  A producer that fills them in place (the HBM-resident feature store gathers a batch
        straight into them, `ResidentFeatures.batch(ids, out=...)`) and then passes the very same tensors to `step` saves the
        device-to-device staging copy of the batch (258 MB per step at batch 64, MSVD-shaped)."""
        if self._graphs is None:
            return None
        st = self._static
        return st['frames'], st['regions'], st['captions'], st['lens']

    def _step_graphs(self, frames, regions, captions, cap_lens, coins, seed, extra_dlogits=None):
        model, ops = self.model, self.model.ops
        hook = extra_dlogits is not None
        if self._graphs is None:
            err = self._capture_agreed(frames, regions, captions, cap_lens, hook)
            if err is not None and self.graph_fallback and self._comm_mode() == 'rccl' and self.world_size > 1:
                # the in-graph RCCL capture was refused somewhere: EVERY rank retries the segmented form (torch.distributed
                # collectives issued by the host between graph segments) -- a rank replaying graphs that hold private-communicator
                # collectives next to ranks issuing host collectives would hang the job
                import warnings
                warnings.warn('capture with in-graph RCCL collectives failed (%s: %s); all ranks retry with host-issued '
                              'torch.distributed collectives between graph segments' % (type(err).__name__, err))
                torch.cuda.synchronize()
                self.comm = 'torch'
                err = self._capture_agreed(frames, regions, captions, cap_lens, hook)
            if err is not None:
                if not self.graph_fallback:
                    raise err
                # opted in: keep the run alive on kernel-by-kernel launches -- on every rank
                import warnings
                warnings.warn('hipGraph capture failed (%s: %s); continuing with eager launches' % (type(err).__name__, err))
                torch.cuda.synchronize()
                self.use_graphs, self._graphs = False, None
                return self._eager_step(frames, regions, captions, cap_lens, coins, seed, counted=True, extra_dlogits=extra_dlogits)
        st = self._static
        if (frames.shape, regions.shape, captions.shape) != (st['frames'].shape, st['regions'].shape, st['captions'].shape):
            # a batch of another shape (the short last batch of an epoch): the captured graphs are for one shape only
            return self._eager_step(frames, regions, captions, cap_lens, coins, seed, counted=True, extra_dlogits=extra_dlogits)
        for k, src in (('frames', frames), ('regions', regions), ('captions', captions), ('lens', cap_lens)):
            if src.data_ptr() != st[k].data_ptr():
                st[k].copy_(src, non_blocking=True)
        self._send_scalars(coins, seed, self._hyper())
        self._works = []
        for g, key in self._graphs:
            g.replay()
            if key == 'hook':
                if getattr(extra_dlogits, 'takes_dst', False):
                    extra_dlogits(*self._hook_sv, dst=st['extra'])         # writes the static buffer itself: no staging copy
                else:
                    st['extra'].copy_(extra_dlogits(*self._hook_sv))
            elif key is not None:
                self._allreduce(key)
        if not self._adam_in_graph:
            self._reduce_guard()
            for w in self._works:
                w.wait()
            # same arithmetic as the captured Adam launch (bias corrections read from the device word the host just wrote),
            # so a segmented step is bit-identical to the single-graph step
            self._adam(self.t, hyper=st['hyper'])
        # the loss lives in the graphs' static memory: hand out a copy, so losses kept across steps do not alias
        return self._loss.clone()
