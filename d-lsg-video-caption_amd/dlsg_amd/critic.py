"""The `DiscV2` critic (models/model.py:110-168, models/layer.py:661-715) and its WGAN-GP update (run_gun.py:339-398) as an explicit
launch schedule over the kernel interface (`ops`, hip.py) -- what engine.py is for the generator.  No autograd inside.

A critic update differentiates the critic twice (the gradient penalty takes d(score)/d(input) with create_graph=True and the loss
backward runs through that gradient, run_gun.py:362-371).  Here that is four passes over saved activations:

  F   forward of the three caption sets [real | fake | mixed] as ONE 3B-caption batch (the Conv1d(V -> 512, k = 1) is linear, so
      the sets share one logit projection and the mixed set is mixed AFTER it);
  B1  backward of sum(mixed scores) through the mixed captions only, inputs only:  g = d(sum mixed) / d(mixed projection);
      |d mixed / d mixed_captions|^2 = sum_l g_l (W W^T) g_l^T  (a 512 x 512 Gram matrix stands in for the (B, L, V) gradient);
  T   with v = 10 d(penalty)/dg: the derivative of every block of F and of B1 ALONG v (mixed captions only).  By the symmetry of
      second derivatives, d/d(theta) [v . dS/dh] enters the third pass as: the derivative of a block's forward output = the
      tangent the next block consumes; the derivative of its backward's input gradient = an extra cotangent on that input; the
      derivative of its backward's parameter gradient = an extra parameter gradient.  For a linear layer y = x W^T the last is
      c1^T xdot (c1 = B1's cotangent of y, xdot = the tangent of x), so tangents live in a FOURTH caption slot of the activation
      buffers and B1's cotangents in the fourth slot of the cotangent buffers: every weight gradient is one TN product over all
      four slots;
  B2  backward of mean(fake) - mean(real) over the 3B captions, with T's extra cotangents added on the mixed captions.

All activations are batch-major (caption, word, channel): per-caption kernels see dense (L, C) blocks, products see contiguous
row ranges per caption set.  Dropout (train mode) is the stateless counter mask of the generator path, recomputed in every pass.
"""
import math

import torch

from .hip import GEMM_NT, GEMM_NN, GEMM_TN, F_ACCUM, F_TANH, F_FORCE64, F_FORCE128

C = 512              # DiscV2.dim and every internal width (models/model.py:113, layer.py:665-683)
PW = 1024            # proposals enter through Linear(1024, 512) (layer.py:666)
P_DROP = 0.3         # every dropout of the critic (models/model.py:125,128; layer.py:680; sublayer.py:186)
# dropout sites (two consecutive numbers where the two proposal heads run side by side)
SITE_LSTM, SITE_SA, SITE_PSL, SITE_TSUM = 1, 2, 3, 5
HEADS = ('obj_psl_score', 'motion_psl_score')          # head 0 scores the object proposals, head 1 the motion proposals


class _Ws(object):
    """buffers of one (B, L, V, slots) shape; slot = a block of B captions: [real | fake | mixed | tangent / B1 cotangent]"""

    def __init__(self, dev, B, L, V, T, P, slots):
        self.B, self.L, self.V, self.T, self.P, self.slots = B, L, V, T, P, slots
        self.dev = dev
        self._b = {}

    def get(self, name, *shape, dtype=torch.float32, zero=False):
        t = self._b.get(name)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(*shape, dtype=dtype, device=self.dev)
            self._b[name] = t
        assert tuple(t.shape) == tuple(shape), (name, t.shape, shape)
        return t


class CriticEngine(object):
    """schedules of a `DiscV2` whose parameters live in a flat arena (`D.flatten_parameters_()`)"""

    def __init__(self, D):
        self.D = D
        self._wss = {}
        self.seed_counter = 0

    # ------------------------------------------------------------------ parameters
    def _params(self):
        D = self.D
        D.flatten_parameters_()
        if getattr(self, '_p_arena', None) is D._flat:
            return self._p, self._g
        P = dict(D.named_parameters())
        G = D.grad_views()
        p, g = {}, {}
        for src, dst in ((P, p), (G, g)):
            dst['Wc'] = src['block.0.res_block.1.weight'].view(C, 3 * C)            # column 3 c + k, as Conv1d stores it
            dst['bc'] = src['block.0.res_block.1.bias']
            dst['Wvoc'] = src['conv1d.weight'].view(C, -1)
            dst['bvoc'] = src['conv1d.bias']
            dst['W_ih'], dst['W_hh'] = src['lstm.weight_ih_l0'], src['lstm.weight_hh_l0']
            dst['b_ih'], dst['b_hh'] = src['lstm.bias_ih_l0'], src['lstm.bias_hh_l0']
            dst['ln_g'], dst['ln_b'] = src['layer_norm.weight'], src['layer_norm.bias']
            dst['WK'], dst['WQ'], dst['WV'] = src['att.K.weight'], src['att.Q.weight'], src['att.V.weight']
            dst['Wo'] = src['att.output_layer.0.weight']
            dst['an_g'], dst['an_b'] = src['att_norm.1.weight'], src['att_norm.1.bias']
            dst['theta'] = src['text_sum.theta']
            dst['ts_g'], dst['ts_b'] = src['text_sum.out_norm.1.weight'], src['text_sum.out_norm.1.bias']
            dst['fusion'] = src['fusion']
            for k, tail in (('wc', 'psl_scorer.classify.weight'), ('bcl', 'psl_scorer.classify.bias'),
                            ('Wv', 'psl_scorer.visual_embed.0.weight'), ('bv', 'psl_scorer.visual_embed.0.bias'),
                            ('Ws', 'psl_scorer.sent_embed.0.weight'), ('bs', 'psl_scorer.sent_embed.0.bias'),
                            ('We', 'psl_embed.0.weight'), ('be', 'psl_embed.0.bias'),
                            ('eg', 'psl_embed.2.weight'), ('eb', 'psl_embed.2.bias'),
                            ('pn_g', 'psl_norm.1.weight'), ('pn_b', 'psl_norm.1.bias'),
                            ('Wa', 'att_norm.0.weight'), ('ba', 'att_norm.0.bias'),
                            ('aa_g', 'att_norm.2.weight'), ('aa_b', 'att_norm.2.bias')):
                dst[k] = [src['%s.%s' % (h, tail)] for h in HEADS]
            # K, Q, V weights are consecutive 512 x 512 blocks of the arena: one (1536, 512) operand
            k_, q_, v_ = dst['WK'], dst['WQ'], dst['WV']
            assert q_.data_ptr() == k_.data_ptr() + 4 * C * C and v_.data_ptr() == q_.data_ptr() + 4 * C * C
            dst['Wkqv'] = torch.as_strided(k_, (3 * C, C), (C, 1))
        self._p, self._g, self._p_arena = p, g, D._flat
        return p, g

    def next_seed(self):
        self.seed_counter += 1
        return (0x2545F4914F6CDD1D * self.seed_counter + 0x9E3779B9) & 0xFFFFFFFFFFFF

    def ws(self, dev, B, L, V, slots):
        D = self.D
        T = D.num_top if D.num_psl > D.num_top else D.num_psl
        key = (str(dev), B, L, V, slots)
        w = self._wss.get(key)
        if w is None:
            w = self._wss[key] = _Ws(dev, B, L, V, T, D.num_psl, slots)
        return w

    # ------------------------------------------------------------------ F: captions [0, nf)
    def _drop(self):
        return P_DROP if self.D.training else 0.0

    @torch.no_grad()
    def proposals(self, ws, obj, mot, alpha, smask):
        """what depends on the clips only (not on the captions scored against them): the top-k proposal rows.  Once per batch."""
        ops, D = self.D.ops, self.D
        B, T, P = ws.B, ws.T, ws.P
        ws.psl = (obj.reshape(B * P, PW), mot.reshape(B * P, PW))
        ws.select = D.num_psl > D.num_top
        if ws.select:
            idx = ws.get('idx', 2, B, T, dtype=torch.int64)
            ops.crit_topk(alpha, smask, P, T, idx)

    def _embed_proposals(self, ws, p):
        """psl_embed + visual_embed of both heads (layer.py:692-696,709): they depend on the critic's weights, so once per update"""
        ops = self.D.ops
        B, T, P = ws.B, ws.T, ws.P
        epre, e = ws.get('epre', 2, B * P, C), ws.get('e', 2, B * P, C)
        ops.gemm(GEMM_NT, [(ws.psl[h], p['We'][h], epre[h], p['be'][h]) for h in range(2)])
        ops.cln_fwd([epre[0], epre[1]], p['eg'], p['eb'], [e[0], e[1]], True)
        if ws.select:
            esel = ws.get('esel', 2, B * T, C)
            ops.gather_rows(e.view(2 * B * P, C), ws.get('idx', 2, B, T, dtype=torch.int64).view(-1), esel.view(2 * B * T, C))
        else:
            esel = e
        v = ws.get('v', 2, B * T, C)
        ops.gemm(GEMM_NT, [(esel[h], p['Wv'][h], v[h], p['bv'][h]) for h in range(2)], flags=F_TANH)
        ws.esel, ws.vv = esel, v

    def _bufs(self, ws):
        """named views of the workspace (allocated on first use)"""
        B, L, T, S = ws.B, ws.L, ws.T, ws.slots
        n, ng = S * B, min(S, 3)
        b = ws.__dict__.get('_views')
        if b is not None:
            return b
        g = ws.get
        b = dict(
            h=g('h', ng * B, L, C), x1=g('x1', n, L, C), taps=g('taps', n, L, 3 * C), xin=g('xin', ng * B, L, 4 * C),
            As=g('As', ng * B, L, 4 * C), Hs=g('Hs', n, L, C), Cs=g('Cs', ng * B, L, C), Hprev=g('Hprev', n, L, C),
            y=g('y', n, L, C), KQV=g('KQV', n, L, 3 * C), w=g('w', ng * B, L, L), ctx=g('ctx', n, L, C), out=g('out', n, L, C),
            words=g('words', n, L, C), apre=g('apre', 2, n, L, C), a=g('a', 2, n, L, C), P=g('P', 2, ng * B, L, T),
            wgt=g('wgt', 2, n, T), aggpre=g('aggpre', 2, n, T, C), agg=g('agg', 2, n, T, C), s=g('s', 2, ng * B, T, C),
            adj=g('adj', ng * B, L), u=g('u', ng * B, C), sent=g('sent', ng * B, C), fus=g('fus', n, 2),
            pair=g('pair', 2, ng * B, T), score=g('score', 2, ng * B), both=g('both', ng, 2), outv=g('outv', ng * B),
            # cotangents
            c_fus=g('c_fus', n, 2), c_spre=g('c_spre', 2, n, T, C), c_wgt=g('c_wgt', 2, n, T), c_vcap=g('c_vcap', 2, n, T, C),
            c_agg=g('c_agg', 2, n, T, C), c_aggpre=g('c_aggpre', 2, n, T, C), c_a=g('c_a', 2, n, L, C), de=g('de', 2, n, T, C),
            c_apre=g('c_apre', 2, n, L, C), c_words=g('c_words', 3, n, L, C), c_out=g('c_out', n, L, C), c_ctx=g('c_ctx', n, L, C),
            c_KQV=g('c_KQV', n, L, 3 * C), c_y=g('c_y', n, L, C), c_hs=g('c_hs', n, L, C), DA=g('DA', n, L, 4 * C),
            DH=g('DH', n, L, C), DC=g('DC', n, L, C), c_x1=g('c_x1', n, L, C), c_taps=g('c_taps', n, L, 3 * C),
            c_h=g('c_h', ng * B, L, C),
        )
        ws._views = b
        return b

    def _forward(self, ws, p, nf, seed):
        """critic scores of captions [0, nf) from b['h'][:nf] (their 512-wide projections) -> b['outv'][:nf]"""
        ops, D = self.D.ops, self.D
        B, L, T = ws.B, ws.L, ws.T
        b = self._bufs(ws)
        R = nf * L
        pd = self._drop()
        ng = nf // B
        h, x1, taps = b['h'][:nf], b['x1'][:nf], b['taps'][:nf]
        ops.crit_relu_taps(h, h, p['bc'], 0.3, x1, taps)                       # ResBlock: in-place ReLU feeds the skip too
        ops.gemm(GEMM_NT, [(taps.view(R, 3 * C), p['Wc'], x1.view(R, C))], alpha=0.3, flags=F_ACCUM)
        xin = b['xin'][:nf]
        # (tiles: tools/archive/critic_gemm_probe.py on an MI355X -- 4992 x 2048 x 512: 107 us on 64 x 64 tiles, 124 on the dispatcher's choice)
        ops.gemm(GEMM_NT, [(x1.view(R, C), p['W_ih'], xin.view(R, 4 * C))], flags=F_FORCE64 if R >= 4096 else 0)
        ops.lstm_seq_fwd(xin, p['W_hh'], p['b_ih'], p['b_hh'], b['As'][:nf], b['Hs'][:nf], b['Cs'][:nf], b['Hprev'][:nf])
        y = b['y'][:nf]
        ops.cln_fwd([b['Hs'][:nf].view(R, C)], [p['ln_g']], [p['ln_b']], [y.view(R, C)], False, p_post=pd, site_post=SITE_LSTM, seed=seed)
        KQV = b['KQV'][:nf]
        ops.gemm(GEMM_NT, [(y.view(R, C), p['Wkqv'], KQV.view(R, 3 * C))], flags=F_FORCE128 if R >= 4096 else 0)   # 79 us against 99
        ops.crit_sa_fwd(KQV, ws.smask, b['w'][:nf], b['ctx'][:nf], 1.0 / math.sqrt(D.att.attention_size))
        out, words = b['out'][:nf], b['words'][:nf]
        ops.gemm(GEMM_NT, [(b['ctx'][:nf].view(R, C), p['Wo'], out.view(R, C))])
        ops.cln_fwd([out.view(R, C)], [p['an_g']], [p['an_b']], [words.view(R, C)], True, p_pre=self._sa_drop(), site_pre=SITE_SA, seed=seed)
        apre, a = b['apre'], b['a']
        ops.gemm(GEMM_NT, [(words.view(R, C), p['Wa'][k], apre[k, :nf].view(R, C), p['ba'][k]) for k in range(2)])
        ops.cln_fwd([apre[k, :nf].view(R, C) for k in range(2)], p['aa_g'], p['aa_b'], [a[k, :nf].view(R, C) for k in range(2)], True)
        e2 = [ws.esel[k].view(B, T, C) for k in range(2)]
        ops.crit_pattn_fwd([a[k, :nf] for k in range(2)], e2, ws.smask, [b['P'][k, :nf] for k in range(2)],
                           [b['wgt'][k, :nf] for k in range(2)], [b['aggpre'][k, :nf] for k in range(2)], 1.0 / math.sqrt(C))
        Rt = nf * T
        ops.cln_fwd([b['aggpre'][k, :nf].view(Rt, C) for k in range(2)], p['pn_g'], p['pn_b'],
                    [b['agg'][k, :nf].view(Rt, C) for k in range(2)], True, p_post=pd, site_post=SITE_PSL, seed=seed)
        ops.gemm(GEMM_NT, [(b['agg'][k, :nf].view(Rt, C), p['Ws'][k], b['s'][k, :nf].view(Rt, C), p['bs'][k]) for k in range(2)],
                 flags=F_TANH)
        ops.crit_tsum_fwd(words, p['theta'], p['ts_g'], p['ts_b'], p['fusion'], b['adj'][:nf], b['u'][:nf], b['sent'][:nf],
                          b['fus'][:nf], p=pd, site=SITE_TSUM, seed=seed)
        v2 = [ws.vv[k].view(B, T, C) for k in range(2)]
        ops.crit_score_fwd(v2, [b['s'][k, :nf] for k in range(2)], p['wc'], p['bcl'], [b['wgt'][k, :nf] for k in range(2)],
                           b['fus'][:nf], [b['pair'][k, :nf] for k in range(2)], [b['score'][k, :nf] for k in range(2)],
                           b['both'][:ng], b['outv'][:nf], ng)
        return b['outv'][:nf]

    def _sa_drop(self):
        return float(self.D.att.dropout) if self.D.training else 0.0

    # ------------------------------------------------------------------ B: cotangents of captions [a0, a1) into cotangent slots from c0
    def _backward(self, ws, p, a0, a1, c0, d_out, seed, params=None, acc=None):
        """d_out (a1 - a0): gradient of the loss w.r.t. the critic scores of captions [a0, a1).  Cotangents go to captions
        [c0, c0 + a1 - a0) of the c_* buffers.  params: the gradient dict to write the parameter gradients of the fused kernels into
        (None: inputs only).  acc = (lo, hi) relative to a0: captions whose cotangents already hold the second-order terms."""
        ops, D = self.D.ops, self.D
        B, L, T = ws.B, ws.L, ws.T
        b = self._bufs(ws)
        nb = a1 - a0
        ng = nb // B
        c1 = c0 + nb
        R, Rt = nb * L, nb * T
        pd = self._drop()
        row0, row0t = a0 * L, a0 * T
        sc_a = 1.0 / math.sqrt(C)
        A = lambda name: b[name][a0:a1]                     # activations of these captions
        A2 = lambda name: [b[name][k, a0:a1] for k in range(2)]
        Cc = lambda name: b[name][c0:c1]                    # their cotangents
        C2 = lambda name: [b[name][k, c0:c1] for k in range(2)]
        v2 = [ws.vv[k].view(B, T, C) for k in range(2)]
        e2 = [ws.esel[k].view(B, T, C) for k in range(2)]
        want = params is not None
        part_wc = [ws.get('part_wc', 2, 3 * B, C)[k, :nb] for k in range(2)] if want else None
        dbc = ws.get('dbc', 2) if want else None
        ops.crit_score_bwd(v2, A2('s'), p['wc'], A2('wgt'), A('fus'), A2('pair'), A2('score'), b['both'][a0 // B:a1 // B], d_out, Cc('c_fus'), C2('c_spre'), C2('c_vcap') if want else None, C2('c_wgt'),
                           part_wc, dbc, ng, acc=acc)
        cw = b['c_words']
        ts_part = ws.get('ts_part', 3 * B, 5, C)[:nb] if want else None
        ops.crit_tsum_bwd(A('words'), p['theta'], p['ts_g'], p['ts_b'], p['fusion'], A('adj'), A('u'), A('sent'), A('fus'), Cc('c_fus'),
                          cw[0, c0:c1], ts_part, p=pd, site=SITE_TSUM, seed=seed, row0=a0, acc=acc)
        ops.gemm(GEMM_NN, [(b['c_spre'][k, c0:c1].view(Rt, C), p['Ws'][k], b['c_agg'][k, c0:c1].view(Rt, C)) for k in range(2)])
        ops.cln_bwd([t.view(Rt, C) for t in A2('aggpre')], p['pn_g'], [[t.view(Rt, C) for t in C2('c_agg')]],
                    [t.view(Rt, C) for t in C2('c_aggpre')], None, None, True,
                    p_post=pd, site_post=SITE_PSL, seed=seed, row0=row0t, acc=self._rows(acc, T),
                    defer_ws=self._lnws(ws, 'pn', 0, Rt) if want else None)
        ops.crit_pattn_bwd(A2('a'), e2, ws.smask, A2('P'), C2('c_aggpre'), C2('c_wgt'), C2('c_a'), C2('de') if want else None, sc_a, acc=acc)
        ops.cln_bwd([t.view(R, C) for t in A2('apre')], p['aa_g'], [[t.view(R, C) for t in C2('c_a')]],
                    [t.view(R, C) for t in C2('c_apre')], None, None, True, row0=row0,
                    acc=self._rows(acc, L), defer_ws=self._lnws(ws, 'aa', 0, R) if want else None)
        ops.gemm(GEMM_NN, [(b['c_apre'][k, c0:c1].view(R, C), p['Wa'][k], cw[1 + k, c0:c1].view(R, C)) for k in range(2)])
        ops.cln_bwd([A('out').view(R, C)], [p['an_g']], [[cw[k, c0:c1].view(R, C)] for k in range(3)], [Cc('c_out').view(R, C)],
                    None, None, True, p_pre=self._sa_drop(), site_pre=SITE_SA, seed=seed,
                    row0=row0, acc=self._rows(acc, L), defer_ws=self._lnws(ws, 'an', 0, R) if want else None)
        ops.gemm(GEMM_NN, [(Cc('c_out').view(R, C), p['Wo'], Cc('c_ctx').view(R, C))])
        ops.crit_sa_bwd(A('KQV'), ws.smask, A('w'), Cc('c_ctx'), Cc('c_KQV'), 1.0 / math.sqrt(D.att.attention_size), acc=acc)
        ops.gemm(GEMM_NN, [(Cc('c_KQV').view(R, 3 * C), p['Wkqv'], Cc('c_y').view(R, C))])
        ops.cln_bwd([A('Hs').view(R, C)], [p['ln_g']], [[Cc('c_y').view(R, C)]], [Cc('c_hs').view(R, C)],
                    None, None, False, p_post=pd, site_post=SITE_LSTM, seed=seed, row0=row0,
                    acc=self._rows(acc, L), defer_ws=self._lnws(ws, 'ln', 0, R) if want else None)
        inj = acc is not None
        ops.lstm_seq_bwd(A('As'), A('Cs'), p['W_hh'], Cc('c_hs'), ws.get('dAs_inj', 3 * B, L, 4 * C, zero=True)[:nb] if inj else None,
                         ws.get('dCs_inj', 3 * B, L, C, zero=True)[:nb] if inj else None, Cc('DA'), Cc('DH'), Cc('DC'))
        ops.gemm(GEMM_NN, [(Cc('DA').view(R, 4 * C), p['W_ih'], Cc('c_x1').view(R, C))])
        ops.gemm(GEMM_NN, [(Cc('c_x1').view(R, C), p['Wc'], Cc('c_taps').view(R, 3 * C))], alpha=0.3)
        dst = b['c_h'][a0:a1] if c0 == a0 else ws.get('g', B, L, C)
        ops.crit_relu_taps_bwd(Cc('c_x1'), Cc('c_taps'), A('h'), dst)
        return dst

    @staticmethod
    def _rows(acc, k):
        return None if acc is None else (acc[0] * k, acc[1] * k)

    def _lnws(self, ws, key, level, rows):
        """where a LayerNorm's backward (level 0: the last backward pass over `rows` rows; level 1: the T pass) leaves the
        per-workgroup partial sums of (dgamma, dbeta): (G, 2, partial rows, C).  `_param_grads` sums the rows of both levels in the
        one column-sum launch that also folds the biases -- instead of one small reduction launch behind every LayerNorm kernel"""
        G = 2 if key in ('pn', 'aa') else 1
        t = ws.get('lnws%d_%s' % (level, key), G, 2, self.D.ops.cln_ws_rows(rows, C), C)
        ws.__dict__.setdefault('_lnws_seen', {})[(key, level)] = t
        return t

    # ------------------------------------------------------------------ T: derivative of F and B1 along v, mixed captions
    def _second(self, ws, p, vseed, seed):
        ops, D = self.D.ops, self.D
        B, L, T = ws.B, ws.L, ws.T
        b = self._bufs(ws)
        m0, m1, t0, t1 = 2 * B, 3 * B, 3 * B, 4 * B          # mixed captions; their tangents / B1 cotangents
        R, Rt = B * L, B * T
        pd = self._drop()
        row0, row0t = m0 * L, m0 * T
        M = lambda name: b[name][m0:m1]
        M2 = lambda name: [b[name][k, m0:m1] for k in range(2)]
        Tn = lambda name: b[name][t0:t1]
        T2 = lambda name: [b[name][k, t0:t1] for k in range(2)]
        ops.crit_relu_taps(vseed, M('h'), None, 0.0, Tn('x1'), Tn('taps'))
        ops.gemm(GEMM_NT, [(Tn('taps').view(R, 3 * C), p['Wc'], Tn('x1').view(R, C))], alpha=0.3, flags=F_ACCUM)
        ubar = ws.get('ubar', B, L, 4 * C)
        ops.gemm(GEMM_NT, [(Tn('x1').view(R, C), p['W_ih'], ubar.view(R, 4 * C))])
        dAs, dCs = ws.get('dAs_inj', 3 * B, L, 4 * C, zero=True)[m0:m1], ws.get('dCs_inj', 3 * B, L, C, zero=True)[m0:m1]
        ops.lstm_seq_bwd2(M('As'), M('Cs'), p['W_hh'], Tn('DH'), Tn('DC'), ubar, dAs, dCs, Tn('Hs'), Tn('Hprev'), ws.get('gDC', B, L, C))
        ops.cln_bwd2([M('Hs').view(R, C)], [p['ln_g']], [[Tn('c_y').view(R, C)]], [Tn('Hs').view(R, C)], [M('c_hs').view(R, C)],
                     [Tn('y').view(R, C)], None, False, p_post=pd, site_post=SITE_LSTM, seed=seed, row0=row0,
                     defer_ws=self._lnws(ws, 'ln', 1, R))
        ops.gemm(GEMM_NT, [(Tn('y').view(R, C), p['Wkqv'], Tn('KQV').view(R, 3 * C))])
        ops.crit_sa_bwd2(M('KQV'), ws.smask, M('w'), Tn('c_ctx'), Tn('KQV'), Tn('ctx'), M('c_KQV'), 1.0 / math.sqrt(D.att.attention_size))
        ops.gemm(GEMM_NT, [(Tn('ctx').view(R, C), p['Wo'], Tn('out').view(R, C))])
        cw = b['c_words']
        ops.cln_bwd2([M('out').view(R, C)], [p['an_g']], [[cw[k, t0:t1].view(R, C)] for k in range(3)], [Tn('out').view(R, C)],
                     [M('c_out').view(R, C)], [Tn('words').view(R, C)], None, True, p_pre=self._sa_drop(), site_pre=SITE_SA,
                     seed=seed, row0=row0, defer_ws=self._lnws(ws, 'an', 1, R))
        ops.gemm(GEMM_NT, [(Tn('words').view(R, C), p['Wa'][k], b['apre'][k, t0:t1].view(R, C)) for k in range(2)])
        ops.cln_bwd2([t.view(R, C) for t in M2('apre')], p['aa_g'], [[t.view(R, C) for t in T2('c_a')]], [t.view(R, C) for t in T2('apre')],
                     [t.view(R, C) for t in M2('c_apre')], [t.view(R, C) for t in T2('a')], None, True, row0=row0,
                     defer_ws=self._lnws(ws, 'aa', 1, R))
        e2 = [ws.esel[k].view(B, T, C) for k in range(2)]
        ops.crit_pattn_bwd2(M2('a'), e2, ws.smask, M2('P'), T2('c_aggpre'), T2('c_wgt'), T2('a'), T2('aggpre'), T2('wgt'), M2('c_a'), T2('de'),
                            1.0 / math.sqrt(C))
        ops.cln_bwd2([t.view(Rt, C) for t in M2('aggpre')], p['pn_g'], [[t.view(Rt, C) for t in T2('c_agg')]],
                     [t.view(Rt, C) for t in T2('aggpre')], [t.view(Rt, C) for t in M2('c_aggpre')], [t.view(Rt, C) for t in T2('agg')],
                     None, True, p_post=pd, site_post=SITE_PSL, seed=seed, row0=row0t, defer_ws=self._lnws(ws, 'pn', 1, Rt))
        uspre = ws.get('uspre', 2, B, T, C)
        ops.gemm(GEMM_NT, [(b['agg'][k, t0:t1].view(Rt, C), p['Ws'][k], uspre[k].view(Rt, C)) for k in range(2)])
        ops.crit_tsum_bwd2(M('words'), p['theta'], p['ts_g'], p['ts_b'], p['fusion'], Tn('c_fus'), Tn('words'), Tn('fus'), cw[0, m0:m1],
                           ws.get('ts_part2', B, 5, C), p=pd, site=SITE_TSUM, seed=seed, row0=m0)
        v2 = [ws.vv[k].view(B, T, C) for k in range(2)]
        ops.crit_score_bwd2(v2, M2('s'), p['wc'], p['bcl'], M2('wgt'), M('fus'), M2('pair'), M2('score'), ws.ones_B, [uspre[0], uspre[1]], T2('wgt'), Tn('fus'),
                            M('c_fus'), M2('c_spre'), T2('c_vcap'), M2('c_wgt'), [ws.get('part_wc2', 2, B, C)[k] for k in range(2)],
                            ws.get('dbc2', 2))

    def _colsums(self, ws, cs):
        """column sums (sources, out, copy, scale) in two launches: tall sources (the 4 992-row bias gradients) go out as chunks of
        <= 1280 rows into partial rows first -- a descriptor's columns are summed by ONE workgroup per 64 columns"""
        ops = self.D.ops
        first, second = [], []
        for idx, (srcs, out, out_b, scale) in enumerate(cs):
            if max(t.shape[0] for t in srcs) <= 1536:
                first.append((srcs, out, out_b, scale))
                continue
            parts = [t[r0:r0 + 1280] for t in srcs for r0 in range(0, t.shape[0], 1280)]
            scr = ws.get('cs_part_%d' % idx, len(parts), srcs[0].shape[1])
            first += [([pt], scr[j], None, 1.0) for j, pt in enumerate(parts)]
            second.append(([scr], out, out_b, scale))
        ops.crit_colsum(first)
        if second:
            ops.crit_colsum(second)

    def _tn(self, ws, key, items, alpha=1.0):
        """out = alpha A^T B for every (A (K, M), B (K, N), out) of `items` (same shapes) in ONE launch.  A weight gradient of the
        critic has few output tiles (512 x 512 .. 2048 x 512) and a deep contraction (K = every caption row): the K rows go to
        `ns` groups of the launch, each writing its own slab, folded in a fixed order by slab_reduce (tools/archive/critic_gemm_probe.py on
        an MI355X: 512 x 512 x 6656: 147 us as one group, 50 us with 4; 512 x 1536 x 6656: 151 -> 108 us with 8)"""
        ops = self.D.ops
        A0, B0, _ = items[0]
        K, M, N = A0.shape[0], A0.shape[1], B0.shape[1]
        tiles = ((M + 63) // 64) * ((N + 63) // 64)
        ns = 1 if K < 1024 else (2 if K < 4096 else (8 if tiles >= 128 else 4))
        ns = max(1, min(ns, 16 // len(items)))
        if ns == 1:
            ops.gemm(GEMM_TN, items, alpha=alpha)
            return
        step = ((K + ns - 1) // ns + 31) // 32 * 32
        bounds = [(k, min(K, k + step)) for k in range(0, K, step)]
        slabs = ws.get('tn_' + key, len(items), len(bounds), M, N)
        ops.gemm(GEMM_TN, [(A[k0:k1], Bm[k0:k1], slabs[j, i]) for j, (A, Bm, _) in enumerate(items) for i, (k0, k1) in enumerate(bounds)],
                 alpha=alpha)
        ws.reduces += [(slabs[j], out) for j, (_, _, out) in enumerate(items)]

    # ------------------------------------------------------------------ parameter gradients after the last backward pass
    def _param_grads(self, ws, p, g, logits_tm, ids, eps):
        """Weight gradients as TN products over every caption slot (a critic update: the 3B captions of B2 + the tangent /
        B1-cotangent slot; a first-order backward: the one slot), biases and the fused kernels' partials as column sums.  Writes
        every entry of the gradient arena."""
        ops = self.D.ops
        B, L, T, P, V, S = ws.B, ws.L, ws.T, ws.P, ws.V, ws.slots
        second = S == 4
        b = self._bufs(ws)
        na, npr = S * B, min(S, 3) * B                       # captions in the products / captions that carry the loss
        Ra, Rp, Rta, Rtp = na * L, npr * L, na * T, npr * T
        f2 = lambda t, rows: t.reshape(rows, t.shape[-1])
        # ---- the caption trunk (the K-split products leave slabs: folded by ONE launch below)
        tn = self._tn
        ws.reduces = []
        tn(ws, 'Wc', [(f2(b['c_x1'], Ra), f2(b['taps'], Ra), g['Wc'])], alpha=0.3)
        tn(ws, 'Wl', [(f2(b['DA'], Ra), f2(b['x1'], Ra), g['W_ih']), (f2(b['DA'], Ra), f2(b['Hprev'], Ra), g['W_hh'])])
        tn(ws, 'Wkqv', [(f2(b['c_KQV'], Ra), f2(b['y'], Ra), g['Wkqv'])])
        tn(ws, 'Wsq', [(f2(b['c_out'], Ra), f2(b['ctx'], Ra), g['Wo'])] +
           [(f2(b['c_apre'][k], Ra), f2(b['words'], Ra), g['Wa'][k]) for k in range(2)])
        tn(ws, 'Ws', [(f2(b['c_spre'][k], Rta), f2(b['agg'][k], Rta), g['Ws'][k]) for k in range(2)])
        ops.crit_reduce(ws.reduces)
        ws.reduces = []
        # ---- the proposal side: cotangents of e_sel and v summed over the caption slots
        c_vpre, c_esel = ws.get('c_vpre', 2, B * T, C), ws.get('c_esel', 2, B * T, C)
        ops.crit_reduce([(b[name][k].view(S, B * T, C), dst[k]) for k in range(2) for name, dst in (('c_vcap', c_vpre), ('de', c_esel))])
        ops.gemm(GEMM_NN, [(c_vpre[k], p['Wv'][k], c_esel[k]) for k in range(2)], flags=F_ACCUM)
        ops.gemm(GEMM_TN, [(c_vpre[k], ws.esel[k], g['Wv'][k]) for k in range(2)])
        if ws.select:
            c_e = ws.get('c_e', 2, B * P, C)
            ops.crit_unselect(c_esel.view(2 * B * T, C), ws.get('idx', 2, B, T, dtype=torch.int64).view(-1), c_e.view(2 * B * P, C), P)
        else:
            c_e = c_esel
        c_epre = ws.get('c_epre', 2, B * P, C)
        epre = ws.get('epre', 2, B * P, C)
        ops.cln_bwd([epre[0], epre[1]], p['eg'], [[c_e[0], c_e[1]]], [c_epre[0], c_epre[1]], g['eg'], g['eb'], True)
        ops.gemm(GEMM_TN, [(c_epre[k], ws.psl[k], g['We'][k]) for k in range(2)])
        # ---- vocabulary projection: fake logits (TN product); in an update also the real ids (scatter) and the penalty's Gram matrix
        dhr, dhf = (ws.get('dhr', B, L, C) if second else None), ws.get('dhf_tm', L, B, C)
        ops.crit_embed_mix_bwd(b['c_h'].view(min(S, 3), B, L, C), eps, dhr, dhf)
        tn(ws, 'Wvoc', [(dhf.view(L * B, C), logits_tm.reshape(L * B, V), g['Wvoc'])])
        if second:
            Mg = ws.get('Mg', C, C)
            tn(ws, 'Mg', [(ws.get('gsc', B, L, C).view(B * L, C), ws.get('g', B, L, C).view(B * L, C), Mg)])
        if ws.reduces:
            ops.crit_reduce(ws.reduces)
        if second:
            ops.gemm(GEMM_NN, [(Mg, p['Wvoc'], g['Wvoc'])], alpha=2.0, flags=F_ACCUM)
            ops.crit_vocab_scatter(dhr, ids, g['Wvoc'])
        # ---- biases and the fused kernels' per-caption partials: ONE launch of column sums (a descriptor: sources, out, copy, scale)
        cs = []
        cs.append(([dhf.view(L * B, C)] + ([dhr.view(B * L, C)] if second else []), g['bvoc'], None, 1.0))
        cs.append(([f2(b['DA'][:npr], Rp)], g['b_ih'], g['b_hh'], 1.0))
        cs.append(([f2(b['c_x1'][:npr], Rp)], g['bc'], None, 0.3))
        for k in range(2):
            cs.append(([f2(b['c_apre'][k, :npr], Rp)], g['ba'][k], None, 1.0))
            cs.append(([f2(b['c_spre'][k, :npr], Rtp)], g['bs'][k], None, 1.0))
            cs.append(([c_vpre[k]], g['bv'][k], None, 1.0))
            cs.append(([c_epre[k]], g['be'][k], None, 1.0))
        # the last pass's partials over its captions (+ the T pass's over the mixed captions)
        tp = ws.get('ts_part', 3 * B, 5, C)[:npr]
        tp2 = ws.get('ts_part2', B, 5, C) if second else None
        for j, dst in enumerate((g['theta'].view(-1), g['ts_g'], g['ts_b'], g['fusion'][0], g['fusion'][1])):
            cs.append(([tp[:, j]] + ([tp2[:, j]] if second else []), dst, None, 1.0))
        pw = ws.get('part_wc', 2, 3 * B, C)
        pw2 = ws.get('part_wc2', 2, B, C) if second else None
        dbc = ws.get('dbc', 2)
        dbc2 = ws.get('dbc2', 2) if second else None
        for k in range(2):
            cs.append(([pw[k, :npr]] + ([pw2[k]] if second else []), g['wc'][k].view(-1), None, 1.0))
            cs.append(([dbc[k:k + 1].view(1, 1)] + ([dbc2[k:k + 1].view(1, 1)] if second else []), g['bcl'][k].view(-1), None, 1.0))
        # the LayerNorms' (dgamma, dbeta): rows of the partials the backward kernels left (+ the T pass's second-order dgamma)
        for key, dg, db in (('ln', [g['ln_g']], [g['ln_b']]), ('an', [g['an_g']], [g['an_b']]), ('aa', g['aa_g'], g['aa_b']),
                            ('pn', g['pn_g'], g['pn_b'])):
            w0 = ws._lnws_seen[(key, 0)]
            w1 = ws._lnws_seen.get((key, 1)) if second else None
            for gi in range(len(dg)):
                cs.append(([w0[gi, 0]] + ([w1[gi, 0]] if w1 is not None else []), dg[gi].view(-1), None, 1.0))
                cs.append(([w0[gi, 1]], db[gi].view(-1), None, 1.0))
        self._colsums(ws, cs)

    # ------------------------------------------------------------------ one critic update (run_gun.py:343-381)
    @torch.no_grad()
    def update_gradients(self, ws, captions, logits_tm, eps, seed):
        """F, B1, T, B2 and the parameter gradients of loss_D = mean(fake) - mean(real) + 10 penalty for the batch whose clip side
        `proposals()` prepared.  captions (B, L) int64, logits_tm (L, B, V) generator logits, eps (B).  Fills the gradient arena,
        returns stats (8): [loss_D, mean real, mean fake, penalty, Wasserstein estimate, ...]."""
        ops, D = self.D.ops, self.D
        p, g = self._params()
        B, L, V = ws.B, ws.L, ws.V
        b = self._bufs(ws)
        self._embed_proposals(ws, p)
        proj = ws.get('proj_tm', L, B, C)
        ops.gemm(GEMM_NT, [(logits_tm.reshape(L * B, V), p['Wvoc'], proj.view(L * B, C))])
        ops.crit_embed_mix(proj, captions, p['Wvoc'], p['bvoc'], eps, b['h'].view(3, B, L, C))
        out = self._forward(ws, p, 3 * B, seed)
        # B1: d(sum of the mixed scores) / d(mixed projection)
        gbuf = self._backward(ws, p, 2 * B, 3 * B, 3 * B, ws.ones_B, seed)
        Gm, gG = ws.get('gram', C, C), ws.get('gG', B, L, C)
        ops.gemm(GEMM_NT, [(p['Wvoc'], p['Wvoc'], Gm)])
        ops.gemm(GEMM_NT, [(gbuf.view(B * L, C), Gm, gG.view(B * L, C))])
        stats, vseed = ws.get('stats', 8), ws.get('vseed', B, L, C)
        ops.crit_gp(gbuf, gG, out, stats, vseed, ws.get('gsc', B, L, C))
        self._second(ws, p, vseed, seed)
        self._backward(ws, p, 0, 3 * B, 0, ws.d_out3, seed, params=g, acc=(2 * B, 3 * B))
        self._param_grads(ws, p, g, logits_tm, captions, eps)
        return stats

    def prepare(self, dev, B, L, V, smask, slots):
        ws = self.ws(dev, B, L, V, slots)
        ws.smask = smask
        if not hasattr(ws, 'ones_B'):
            ws.ones_B = torch.ones(B, dtype=torch.float32, device=dev)
            ws.d_out3 = torch.cat([torch.full((B,), -1.0 / B), torch.full((B,), 1.0 / B), torch.zeros(B)]).to(dev)
            ws.d_outG = torch.full((B,), -1.0 / B, dtype=torch.float32, device=dev)
        return ws

    # ------------------------------------------------------------------ first-order use: scores and their gradient
    @torch.no_grad()
    def score(self, ws, logits_tm, seed):
        """critic scores (B,) of the captions whose logits (L, B, V) are given, against the clips `proposals()` prepared"""
        ops = self.D.ops
        p, g = self._params()
        B, L, V = ws.B, ws.L, ws.V
        b = self._bufs(ws)
        self.seed_last = seed
        self._embed_proposals(ws, p)
        proj = ws.get('proj_tm', L, B, C)
        ops.gemm(GEMM_NT, [(logits_tm.reshape(L * B, V), p['Wvoc'], proj.view(L * B, C))])
        ops.crit_embed_mix(proj, None, p['Wvoc'], p['bvoc'], None, b['h'].view(1, B, L, C))
        return self._forward(ws, p, B, seed)

    @torch.no_grad()
    def score_backward(self, ws, logits_tm, d_out, seed, dlogits_tm, scale=1.0, params=True):
        """backward of `score`: d_out (B,) -> dlogits_tm (L, B, V) (None: left to `dlogits`) and, with params, the gradient arena.
        Returns the gradient w.r.t. the 512-wide projection, time-major (L, B, 512)."""
        ops = self.D.ops
        p, g = self._params()
        B, L = ws.B, ws.L
        ch = self._backward(ws, p, 0, B, 0, d_out, seed, params=g if params else None)
        if params:
            self._param_grads(ws, p, g, logits_tm, None, None)
            dhf = ws.get('dhf_tm', L, B, C)
        else:
            dhf = ws.get('dhf_tm', L, B, C)
            ops.crit_embed_mix_bwd(ch.view(1, B, L, C), None, None, dhf)
        if dlogits_tm is not None:
            self.dlogits(ws, dhf, dlogits_tm, scale)
        return dhf

    @torch.no_grad()
    def dlogits(self, ws, dhf, dlogits_tm, scale=1.0):
        """dlogits_tm (L, B, V) = scale * dhf (L, B, 512) Wvoc: the gradient reaches the generator's logits (run_gun.py:214-231)"""
        p, _ = self._params()
        self.D.ops.gemm(GEMM_NN, [(dhf.view(ws.L * ws.B, C), p['Wvoc'], dlogits_tm.view(ws.L * ws.B, ws.V))], alpha=scale)
