"""TEST INFRASTRUCTURE ONLY: CPU emulation of the critic-schedule kernels (csrc/critic_sched.hip, csrc/critic_lstm.hip) behind
`dlsg_amd/critic.py`, mixed into tests/emul_ops.EmulOps.

Every fused block is written as a pure function `f` (forward) and `b` (its vector-Jacobian product); the third level -- what the
WGAN-GP update needs of a block (run_gun.py:362-371) -- is the DIRECTIONAL DERIVATIVE of (f, b) along a tangent U of the block's
input, taken here by `torch.func.jvp`.  (The penalty's second-order term is d/d(theta) of U . d(score)/d(input); by the symmetry
of second derivatives the cotangents it sends into a block's inputs and parameters are the derivatives of the block's own
backward outputs along U, and what it hands to the next block is the derivative of the block's forward output.)  The HIP kernels
carry the hand-differentiated lines; the GPU tests hold them to these emulations, and tests/test_critic_engine.py holds the whole
schedule to torch autograd's double backward of the oracle."""
import math

import numpy as np
import torch
from torch.func import jvp


def _m(seed, site, rows, n, p, row0, like):
    if p <= 0.0:
        return None
    from emul_ops import _mask
    return _mask(seed, site, rows, n, p, row0).to(like.dtype)


# ---------------------------------------------------------------------------------------------- (tanh +) LayerNorm with dropouts
def cln_f(x, gamma, beta, pre_tanh, eps, mpre, mpost):
    z = x if mpre is None else x * mpre
    t = torch.tanh(z) if pre_tanh else z
    mu = t.mean(1, keepdim=True)
    r = (((t - mu) ** 2).mean(1, keepdim=True) + eps).rsqrt()
    y = (t - mu) * r * gamma + beta
    return y if mpost is None else y * mpost


def cln_b(x, gamma, dy, pre_tanh, eps, mpre, mpost):
    """-> (dx, dgamma, dbeta)"""
    z = x if mpre is None else x * mpre
    t = torch.tanh(z) if pre_tanh else z
    mu = t.mean(1, keepdim=True)
    r = (((t - mu) ** 2).mean(1, keepdim=True) + eps).rsqrt()
    n = (t - mu) * r
    d = dy if mpost is None else dy * mpost
    a = d * gamma
    dt = r * (a - a.mean(1, keepdim=True) - n * (a * n).mean(1, keepdim=True))
    dz = dt * (1 - t * t) if pre_tanh else dt
    dx = dz if mpre is None else dz * mpre
    return dx, (d * n).sum(0), d.sum(0)


# ---------------------------------------------------------------------------------------------- masked 26 x 26 self-attention core
def _full_mask(smask, n):
    B = smask.shape[0]
    m = (smask.unsqueeze(2) * smask.unsqueeze(1)) > 0           # (B,L,L): run_gun.py:164-166
    return m.repeat(n // B, 1, 1)


def sa_f(K, Q, V, m, scale):
    lg = (K @ Q.transpose(1, 2)) * scale
    lg = torch.where(m, lg, torch.full_like(lg, -9e15))
    w = torch.softmax(lg, 2)
    return w, w @ V


def sa_b(K, Q, V, m, w, dctx, gw, scale):
    dw = dctx @ V.transpose(1, 2)
    if gw is not None:
        dw = dw + gw
    dV = w.transpose(1, 2) @ dctx
    dlg = w * (dw - (w * dw).sum(2, keepdim=True))
    dlg = torch.where(m, dlg, torch.zeros_like(dlg))
    return scale * (dlg @ Q), scale * (dlg.transpose(1, 2) @ K), dV


# ---------------------------------------------------------------------------------------------- PSLScore2's word -> proposal graph
def pattn_f(a, e, sm, scale):
    """a (n,L,C), e (n,T,C) (already repeated per caption), sm (n,L,1) -> P (n,L,T), wgt (n,T), aggpre (n,T,C)"""
    S = (a @ e.transpose(1, 2)) * scale
    P = torch.softmax(S, 1)
    adj = P * sm
    return P, adj.sum(1), adj.transpose(1, 2) @ a


def pattn_b(a, e, sm, P, d_agg, d_wgt, scale):
    adj = P * sm
    dadj = a @ d_agg.transpose(1, 2) + d_wgt.unsqueeze(1)
    dP = dadj * sm
    dS = P * (dP - (P * dP).sum(1, keepdim=True))
    da = adj @ d_agg + scale * (dS @ e)
    de = scale * (dS.transpose(1, 2) @ a)
    return da, de


# ---------------------------------------------------------------------------------------------- text summary + fusion weights
def tsum_f(words, theta, gamma, beta, fusion, eps, mpost):
    lg = words @ theta                                           # (n,L)
    adj = torch.softmax(lg, 1)
    u = (adj.unsqueeze(1) @ words).squeeze(1)                    # (n,C)
    sent = cln_f(u, gamma, beta, True, eps, None, mpost)
    fus = torch.softmax(sent @ fusion.t(), 1)                    # (n,2)
    return adj, u, sent, fus


def tsum_b(words, theta, gamma, fusion, adj, u, sent, fus, d_fus, eps, mpost):
    """-> dwords (n,L,C), part (n,5,C) = per-caption [dtheta, dgamma, dbeta, dfusion_0, dfusion_1]"""
    dfl = fus * (d_fus - (fus * d_fus).sum(1, keepdim=True))     # (n,2)
    dsent = dfl @ fusion
    dfusion = dfl.unsqueeze(2) * sent.unsqueeze(1)               # (n,2,C)
    # LayerNorm backward per row, parameter partials per row
    t = torch.tanh(u)
    mu = t.mean(1, keepdim=True)
    r = (((t - mu) ** 2).mean(1, keepdim=True) + eps).rsqrt()
    nn_ = (t - mu) * r
    d = dsent if mpost is None else dsent * mpost
    a = d * gamma
    dt = r * (a - a.mean(1, keepdim=True) - nn_ * (a * nn_).mean(1, keepdim=True))
    du = dt * (1 - t * t)
    dadj = (words @ du.unsqueeze(2)).squeeze(2)                  # (n,L)
    dlg = adj * (dadj - (adj * dadj).sum(1, keepdim=True))
    dwords = adj.unsqueeze(2) * du.unsqueeze(1) + dlg.unsqueeze(2) * theta
    dtheta = (dlg.unsqueeze(2) * words).sum(1)
    part = torch.stack([dtheta, d * nn_, d, dfusion[:, 0], dfusion[:, 1]], 1)
    return dwords, part


# ---------------------------------------------------------------------------------------------- pair scores -> critic output
def score_f(v, s, wc, bc, wgt, fus, ng):
    """v (2,B,T,C), s (2,n,T,C), wc (2,C), bc (2), wgt (2,n,T), fus (n,2) -> pair (2,n,T), score (2,n), both (2,ng), out (n)"""
    n = s.shape[1]
    B = n // ng
    vr = v.repeat(1, ng, 1, 1)
    pair = (vr * s * wc.view(2, 1, 1, -1)).sum(3) + bc.view(2, 1, 1)
    score = (pair * wgt).sum(2) / wgt.sum(2)
    both = score.view(2, ng, B).mean(2)
    out = (both.repeat_interleave(B, 1) * fus.t()).sum(0)
    return pair, score, both, out


def score_b(v, s, wc, wgt, fus, pair, score, both, d_out, ng):
    """-> d_fus (n,2), c_spre (2,n,T,C), c_vpre per caption (2,n,T,C), d_wgt (2,n,T), part_wc (2,n,C), dbc (2)"""
    n = s.shape[1]
    B = n // ng
    vr = v.repeat(1, ng, 1, 1)
    d_fus = (d_out.unsqueeze(0) * both.repeat_interleave(B, 1)).t()
    d_both = (d_out.unsqueeze(0) * fus.t()).view(2, ng, B).sum(2)
    d_score = (d_both / B).repeat_interleave(B, 1)                                  # (2,n)
    wsum = wgt.sum(2, keepdim=True)
    d_pair = d_score.unsqueeze(2) * wgt / wsum
    d_wgt = d_score.unsqueeze(2) * (pair - score.unsqueeze(2)) / wsum
    w4 = wc.view(2, 1, 1, -1)
    c_spre = d_pair.unsqueeze(3) * vr * w4 * (1 - s * s)
    c_vpre = d_pair.unsqueeze(3) * s * w4 * (1 - vr * vr)
    part_wc = (d_pair.unsqueeze(3) * vr * s).sum(2)
    return d_fus, c_spre, c_vpre, d_wgt, part_wc, d_both.sum(1)


class CriticEmul(object):
    """mixin of tests/emul_ops.EmulOps: the critic-schedule entry points of dlsg_amd.hip.HipOps"""

    # ---- vocabulary projection glue
    def crit_embed_mix(self, proj_tm, ids, W, bias, eps, h):
        hf = proj_tm.transpose(0, 1) + bias
        if h.shape[0] == 1:
            h[0].copy_(hf)
            return
        hr = W.t()[ids] + bias
        e = eps.view(-1, 1, 1)
        h[0].copy_(hr); h[1].copy_(hf); h[2].copy_(e * hr + (1 - e) * hf)

    def crit_embed_mix_bwd(self, ch, eps, dhr, dhf_tm):
        if ch.shape[0] == 1:
            dhf_tm.copy_(ch[0].transpose(0, 1))
            return
        e = eps.view(-1, 1, 1)
        dhr.copy_(ch[0] + e * ch[2])
        dhf_tm.copy_((ch[1] + (1 - e) * ch[2]).transpose(0, 1))

    def crit_vocab_scatter(self, dhr, ids, dW):
        dW.t().index_add_(0, ids.reshape(-1), dhr.reshape(-1, dhr.shape[-1]))

    # ---- ResBlock head: in-place ReLU + the three conv taps, interleaved (column 3 c + k) as Conv1d's weight is stored
    def crit_relu_taps(self, x, ref, bias, bias_scale, y, taps):
        z = x * (ref > 0).to(x.dtype)
        y.copy_(z if bias is None else z + bias_scale * bias)
        zp = torch.nn.functional.pad(z, (0, 0, 1, 1))
        taps.copy_(torch.stack([zp[:, :-2], zp[:, 1:-1], zp[:, 2:]], dim=3).reshape(taps.shape))

    def crit_relu_taps_bwd(self, dy, dtaps, ref, dx):
        n, L, C = dy.shape
        d = torch.nn.functional.pad(dtaps.view(n, L, C, 3), (0, 0, 0, 0, 1, 1))     # pad the word axis
        acc = d[:, 2:, :, 0] + d[:, 1:-1, :, 1] + d[:, :-2, :, 2]
        dx.copy_((dy + acc) * (ref > 0).to(dy.dtype))

    # ---- whole-sequence LSTM, batch-major (n, L, .) arrays
    def lstm_seq_supported(self, L, n, H):
        return True

    def lstm_seq_fwd(self, xin, W, b_ih, b_hh, As, Hs, Cs, Hprev):
        n, L, G = xin.shape
        h = xin.new_zeros(n, G // 4); c = xin.new_zeros(n, G // 4)
        for t in range(L):
            Hprev[:, t].copy_(h)
            a = xin[:, t] + b_ih + b_hh + h @ W.t()
            As[:, t].copy_(a)
            i, f, g, o, c, tc, q = self._cell(a, c)
            h = o * tc
            Hs[:, t].copy_(h); Cs[:, t].copy_(c)

    def lstm_seq_bwd(self, As, Cs, W, dHs, dAs, dCs, DA, DH, DC):
        n, L, G = As.shape
        H = G // 4
        r = As.new_zeros(n, H); s = As.new_zeros(n, H)
        for t in range(L - 1, -1, -1):
            dh = dHs[:, t] + r
            dc = s + (dCs[:, t] if dCs is not None else 0)
            DH[:, t].copy_(dh); DC[:, t].copy_(dc)
            cp = Cs[:, t - 1] if t else torch.zeros_like(dh)
            da = torch.empty(n, G, dtype=As.dtype); s = torch.empty_like(dh)
            self.lstm_cell_bwd(As[:, t], cp, dh, dc, da, s)
            if dAs is not None:
                da = da + dAs[:, t]
            DA[:, t].copy_(da)
            r = da @ W

    def lstm_seq_bwd2(self, As, Cs, W, DH, DC, ubar, gA, gC, gDH, gDHprev, gDC):
        n, L, G = As.shape
        H = G // 4
        gdh = As.new_zeros(n, H); gdc = None
        gC.zero_()
        for t in range(L):
            gDHprev[:, t].copy_(gdh)
            u = ubar[:, t] + gdh @ W.t()
            ga = torch.empty(n, G, dtype=As.dtype); gc = torch.empty(n, H, dtype=As.dtype)
            gdh = torch.empty(n, H, dtype=As.dtype); gdc_new = torch.empty(n, H, dtype=As.dtype)
            self.lstm_cell_bwd2(As[:, t], Cs[:, t - 1] if t else None, DH[:, t], DC[:, t], u, gdc, ga, gc, gdh, gdc_new)
            gdc = gdc_new
            gA[:, t].copy_(ga); gDH[:, t].copy_(gdh); gDC[:, t].copy_(gdc)
            if t:
                gC[:, t - 1].copy_(gc)

    # ---- (tanh +) LayerNorm with dropouts; G same-shape blocks (lists of G arrays) with their own gamma / beta per call
    def _cln_masks(self, g, rows, N, like, p_pre, site_pre, p_post, site_post, seed, row0):
        return (_m(seed, site_pre + g, rows, N, p_pre, row0, like), _m(seed, site_post + g, rows, N, p_post, row0, like))

    def cln_fwd(self, x, gamma, beta, y, pre_tanh, eps=1e-5, p_pre=0.0, site_pre=0, p_post=0.0, site_post=0, seed=0, row0=0):
        """x, y, gamma, beta: lists of G arrays ((rows, N) / (N,)); block g uses dropout sites site_* + g; the mask of element
        (r, j) is keyed by (row0 + r) * N + j"""
        for g in range(len(x)):
            mpre, mpost = self._cln_masks(g, x[g].shape[0], x[g].shape[1], x[g], p_pre, site_pre, p_post, site_post, seed, row0)
            y[g].copy_(cln_f(x[g], gamma[g], beta[g], pre_tanh, eps, mpre, mpost))

    def cln_ws_rows(self, rows, N):
        """the emulation leaves ONE row of partials (the HIP kernels: one per workgroup)"""
        return 1

    def cln_bwd(self, x, gamma, dys, dx, dgamma, dbeta, pre_tanh, eps=1e-5, p_pre=0.0, site_pre=0, p_post=0.0, site_post=0, seed=0,
                row0=0, acc=None, extra=None, defer_ws=None):
        """dys: list (<= 3) of lists of G arrays, summed to dy; acc = (lo, hi): rows [lo, hi) of every block are added to dx instead
        of written; dgamma / dbeta: lists of G destinations or None; extra: optional list of G (2, N) arrays added to them"""
        for g in range(len(x)):
            dy = dys[0][g]
            for more in dys[1:]:
                dy = dy + more[g]
            mpre, mpost = self._cln_masks(g, x[g].shape[0], x[g].shape[1], x[g], p_pre, site_pre, p_post, site_post, seed, row0)
            d, dg, db = cln_b(x[g], gamma[g], dy, pre_tanh, eps, mpre, mpost)
            if acc is not None:
                d[acc[0]:acc[1]] += dx[g][acc[0]:acc[1]]
            dx[g].copy_(d)
            if defer_ws is not None:
                defer_ws[g, 0, 0].copy_(dg); defer_ws[g, 1, 0].copy_(db)
            elif dgamma is not None:
                if extra is not None:
                    dg, db = dg + extra[g][0], db + extra[g][1]
                dgamma[g].copy_(dg); dbeta[g].copy_(db)

    def cln_bwd2(self, x, gamma, dys, U, gx, gdy, gpart, pre_tanh, eps=1e-5, p_pre=0.0, site_pre=0, p_post=0.0, site_post=0, seed=0,
                 row0=0, defer_ws=None):
        """gx = d/dU of cln_b's dx, gdy = d/dU of cln_f (the tangent of y), gpart[g] (2, N) = d/dU of (dgamma, dbeta)"""
        for g in range(len(x)):
            dy = dys[0][g]
            for more in dys[1:]:
                dy = dy + more[g]
            dy = dy.detach()
            mpre, mpost = self._cln_masks(g, x[g].shape[0], x[g].shape[1], x[g], p_pre, site_pre, p_post, site_post, seed, row0)
            ga = gamma[g].detach()
            _, ty = jvp(lambda x_: cln_f(x_, ga, torch.zeros_like(ga), pre_tanh, eps, mpre, mpost), (x[g].detach(),), (U[g].detach(),))
            _, tb = jvp(lambda x_: cln_b(x_, ga, dy, pre_tanh, eps, mpre, mpost), (x[g].detach(),), (U[g].detach(),))
            gdy[g].copy_(ty); gx[g].copy_(tb[0])
            if defer_ws is not None:
                defer_ws[g, 0, 0].copy_(tb[1])              # (the dbeta half is zero by construction and is not read)
            else:
                gpart[g][0].copy_(tb[1]); gpart[g][1].copy_(tb[2])

    # ---- masked self-attention core on [K | Q | V] rows
    @staticmethod
    def _kqv(KQV):
        C = KQV.shape[2] // 3
        return KQV[:, :, :C], KQV[:, :, C:2 * C], KQV[:, :, 2 * C:]

    def crit_sa_fwd(self, KQV, smask, w, ctx, scale):
        K, Q, V = self._kqv(KQV)
        ww, c = sa_f(K, Q, V, _full_mask(smask, KQV.shape[0]), scale)
        w.copy_(ww); ctx.copy_(c)

    def crit_sa_bwd(self, KQV, smask, w, dctx, dKQV, scale, acc=None):
        """acc = (lo, hi): the dKQV rows of captions [lo, hi) are added to, not written"""
        K, Q, V = self._kqv(KQV)
        d = torch.cat(sa_b(K, Q, V, _full_mask(smask, KQV.shape[0]), w, dctx, None, scale), 2)
        if acc is not None:
            d[acc[0]:acc[1]] += dKQV[acc[0]:acc[1]]
        dKQV.copy_(d)

    def crit_sa_bwd2(self, KQV, smask, w, dctx, U, Uctx, gKQV, scale):
        """Uctx = d/dU of the forward's ctx; gKQV = d/dU of the backward's dKQV at fixed dctx (w varies with K, Q)"""
        K, Q, V = [t.detach() for t in self._kqv(KQV)]
        UK, UQ, UV = [t.detach() for t in self._kqv(U)]
        m = _full_mask(smask, KQV.shape[0])
        dc = dctx.detach()

        def fb(k, q, v):
            w, ctx = sa_f(k, q, v, m, scale)
            return (ctx,) + sa_b(k, q, v, m, w, dc, None, scale)
        _, t = jvp(fb, (K, Q, V), (UK, UQ, UV))
        Uctx.copy_(t[0]); gKQV.copy_(torch.cat(t[1:], 2))

    # ---- PSLScore2's word -> proposal graph, both heads per call (lists of two arrays; caption i scores the proposals of clip i % B)
    def crit_pattn_fwd(self, a, e, smask, P, wgt, aggpre, scale):
        n = a[0].shape[0]
        sm = smask.repeat(n // smask.shape[0], 1).unsqueeze(2)
        for h in range(2):
            p_, w_, g_ = pattn_f(a[h], e[h].repeat(n // e[h].shape[0], 1, 1), sm, scale)
            P[h].copy_(p_); wgt[h].copy_(w_); aggpre[h].copy_(g_)

    def crit_pattn_bwd(self, a, e, smask, P, d_agg, d_wgt, da, de, scale, acc=None):
        """da[h] (n,L,C): rows of captions in acc are added to; de[h] (n,T,C) per caption (the caller sums the captions of a clip)"""
        n = a[0].shape[0]
        sm = smask.repeat(n // smask.shape[0], 1).unsqueeze(2)
        for h in range(2):
            d, dE = pattn_b(a[h], e[h].repeat(n // e[h].shape[0], 1, 1), sm, P[h], d_agg[h], d_wgt[h], scale)
            if acc is not None:
                d[acc[0]:acc[1]] += da[h][acc[0]:acc[1]]
            da[h].copy_(d)
            if de is not None:
                de[h].copy_(dE)

    def crit_pattn_bwd2(self, a, e, smask, P, d_agg, d_wgt, Ua, Uagg, Uwgt, ga, ge, scale):
        n = a[0].shape[0]
        sm = smask.repeat(n // smask.shape[0], 1).unsqueeze(2)
        for h in range(2):
            eh, dg, dw = e[h].repeat(n // e[h].shape[0], 1, 1).detach(), d_agg[h].detach(), d_wgt[h].detach()

            def fb(a_):
                P_, w_, g_ = pattn_f(a_, eh, sm, scale)
                return (w_, g_) + pattn_b(a_, eh, sm, P_, dg, dw, scale)
            _, t = jvp(fb, (a[h].detach(),), (Ua[h].detach(),))
            Uwgt[h].copy_(t[0]); Uagg[h].copy_(t[1]); ga[h].copy_(t[2]); ge[h].copy_(t[3])

    # ---- text summary (LatentPSL with one node) + fusion softmax
    def crit_tsum_fwd(self, words, theta, gamma, beta, fusion, adj, u, sent, fus, eps=1e-5, p=0.0, site=0, seed=0, row0=0):
        mp = _m(seed, site, words.shape[0], words.shape[2], p, row0, words)
        r = tsum_f(words, theta.reshape(-1), gamma, beta, fusion, eps, mp)
        for dst, src in zip((adj, u, sent, fus), r):
            dst.copy_(src)

    def crit_tsum_bwd(self, words, theta, gamma, beta, fusion, adj, u, sent, fus, d_fus, dwords, part, eps=1e-5, p=0.0, site=0, seed=0,
                      row0=0, acc=None):
        """part (n, 5, C) or None: per-caption partials of [dtheta, dgamma, dbeta, dfusion_0, dfusion_1]"""
        mp = _m(seed, site, words.shape[0], words.shape[2], p, row0, words)
        d, pt = tsum_b(words, theta.reshape(-1), gamma, fusion, adj, u, sent, fus, d_fus, eps, mp)
        if acc is not None:
            d[acc[0]:acc[1]] += dwords[acc[0]:acc[1]]
        dwords.copy_(d)
        if part is not None:
            part.copy_(pt)

    def crit_tsum_bwd2(self, words, theta, gamma, beta, fusion, d_fus, U, Ufus, gwords, gpart, eps=1e-5, p=0.0, site=0, seed=0, row0=0):
        mp = _m(seed, site, words.shape[0], words.shape[2], p, row0, words)
        th, ga, be, fu, df = [t.detach() for t in (theta.reshape(-1), gamma, beta, fusion, d_fus)]

        def fb(w_):
            adj, u, sent, fus = tsum_f(w_, th, ga, be, fu, eps, mp)
            return (fus,) + tsum_b(w_, th, ga, fu, adj, u, sent, fus, df, eps, mp)
        _, t = jvp(fb, (words.detach(),), (U.detach(),))
        Ufus.copy_(t[0]); gwords.copy_(t[1]); gpart.copy_(t[2])

    # ---- pair scores, batch means, critic output (two-element lists per head)
    @staticmethod
    def _st(ts):
        return torch.stack([t.reshape(t.shape) for t in ts])

    def crit_score_fwd(self, v, s, wc, bc, wgt, fus, pair, score, both, out, ng):
        r = score_f(self._st(v), self._st(s), torch.stack([w.reshape(-1) for w in wc]), torch.cat([b.reshape(1) for b in bc]),
                    self._st(wgt), fus, ng)
        for h in range(2):
            pair[h].copy_(r[0][h]); score[h].copy_(r[1][h])
        both.copy_(r[2].t()); out.copy_(r[3])                     # both (ng, 2): [caption set, head]

    def crit_score_bwd(self, v, s, wc, wgt, fus, pair, score, both, d_out, d_fus, c_spre, c_vpre, d_wgt, part_wc, dbc, ng, acc=None):
        """acc = (lo, hi) captions: the d_fus, c_spre, d_wgt rows of those captions are added to, not written.  c_vpre[h] (n,T,C) per
        caption; part_wc[h] (n, C) per-caption partials of classify.weight's gradient, dbc (2) of its bias (None: not wanted)"""
        r = list(score_b(self._st(v), self._st(s), torch.stack([w.reshape(-1) for w in wc]), self._st(wgt), fus, self._st(pair),
                         self._st(score), both.t(), d_out, ng))
        if acc is not None:
            lo, hi = acc
            r[0][lo:hi] += d_fus[lo:hi]
            for h in range(2):
                r[1][h][lo:hi] += c_spre[h][lo:hi]
                r[3][h][lo:hi] += d_wgt[h][lo:hi]
        d_fus.copy_(r[0])
        for h in range(2):
            c_spre[h].copy_(r[1][h]); d_wgt[h].copy_(r[3][h])
            if c_vpre is not None:
                c_vpre[h].copy_(r[2][h])
            if part_wc is not None:
                part_wc[h].copy_(r[4][h])
        if dbc is not None:
            dbc.copy_(r[5])

    def crit_score_bwd2(self, v, s, wc, bc, wgt, fus, pair, score, d_out, Uspre, Uwgt, Ufus, g_fus, g_spre, g_vpre, g_wgt, gpart_wc, g_dbc):
        """one caption group: tangents of s_pre (s = tanh(s_pre)), wgt, fus -> the derivatives of crit_score_bwd's outputs"""
        W = torch.stack([w.reshape(-1) for w in wc]).detach()
        bcv = torch.cat([b.reshape(1) for b in bc]).detach()
        vd, do = self._st(v).detach(), d_out.detach()

        def fb(s_, w_, f_):
            pair, score, both, out = score_f(vd, s_, W, bcv, w_, f_, 1)
            return score_b(vd, s_, W, w_, f_, pair, score, both, do, 1)
        sd = self._st(s).detach()
        _, t = jvp(fb, (sd, self._st(wgt).detach(), fus.detach()),
                   (self._st(Uspre).detach() * (1 - sd * sd), self._st(Uwgt).detach(), Ufus.detach()))
        g_fus.copy_(t[0]); g_dbc.copy_(t[5])
        for h in range(2):
            g_spre[h].copy_(t[1][h]); g_vpre[h].copy_(t[2][h]); g_wgt[h].copy_(t[3][h]); gpart_wc[h].copy_(t[4][h])

    # ---- gradient penalty and the update's loss values
    def crit_gp(self, g, gG, out, stats, vseed, gsc):
        """out (3B): critic scores of [real | fake | mixed]; stats[0:5] = loss_D, mean real, mean fake, penalty, Wasserstein estimate;
        vseed = 10 d(penalty)/dg, gsc = 10 c_b g with c_b = d(penalty)/d(|g_b|_G^2)"""
        B = g.shape[0]
        q = (g * gG).sum((1, 2))
        gn = q.clamp_min(1e-24).sqrt()
        gp = ((gn - 1) ** 2).mean()
        c = torch.where(q > 1e-24, (gn - 1) / (B * gn), torch.zeros_like(gn)).view(B, 1, 1)
        vseed.copy_(20 * c * gG); gsc.copy_(10 * c * g)
        r, f = out[:B].mean(), out[B:2 * B].mean()
        stats[:5].copy_(torch.stack([f - r + 10 * gp, r, f, gp, r - f]))

    # ---- top-k proposals of a clip by attention mass (layer.py:694-696) and the inverse of the gather
    def crit_topk(self, alpha, smask, P, T, idx):
        """idx (2, B, T) int64: rows of the (2 B P, C) proposal embeddings; head 0 = the first P columns of alpha, 1 = the last P"""
        B = alpha.shape[0]
        a = alpha * smask.unsqueeze(2)
        for h, sl in enumerate((slice(0, P), slice(alpha.shape[2] - P, alpha.shape[2]))):
            top = a[:, :, sl].sum(1).topk(T, dim=-1).indices                       # (B,T)
            idx[h].copy_((h * B + torch.arange(B).unsqueeze(1)) * P + top)

    def crit_reduce(self, descs):
        for slabs, out in descs:
            out.copy_(slabs.sum(0).reshape(out.shape))

    def crit_colsum(self, descs):
        for srcs, out, out_b, scale in descs:
            r = srcs[0].sum(0)
            if len(srcs) > 1:
                r = r + srcs[1].sum(0)
            out.copy_((scale * r).reshape(out.shape))
            if out_b is not None:
                out_b.copy_(out)

    def crit_unselect(self, src, idx, dst, per):
        """dst (R, C): row idx[r] = src[r], every other row zero"""
        dst.zero_()
        dst[idx.reshape(-1)] = src.reshape(-1, src.shape[-1])
