"""dlsg_amd/critic.py (the DiscV2 critic and its WGAN-GP update as four explicit passes) on the CPU emulation of the kernels,
against torch autograd:
  * eval mode: the oracle's restatement of the reference (oracle/gan_ref.py, pinned by tests/golden/gan_*.npz), its own
    create_graph double backward, in float64;
  * train mode (dropout): a differentiable composition of the same pure block functions (tests/emul_critic.py) with the engine's
    counter-hash masks, differentiated twice by autograd."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import dlsg_amd
from dlsg_amd import critic as CR
from dlsg_amd.synth import synth_state_dict
from emul_ops import EmulOps
import emul_critic as EC
from helpers import gan_args


def make_case(B=3, L=9, V=37, seed=5, topk=3, P=8):
    args = gan_args(num_topk=topk, num_proposals=P)
    torch.manual_seed(seed)
    D = dlsg_amd.DiscV2(args, V).set_ops(EmulOps())
    D.load_state_dict(synth_state_dict(D.state_dict(), seed + 1))
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(2, L + 1, (B,), generator=g)
    lens[0] = L
    caps = torch.randint(4, V, (B, L), generator=g)
    for b in range(B):
        caps[b, int(lens[b]):] = 0
    logits_tm = torch.randn(L, B, V, generator=g) * 2
    obj = torch.randn(B, P, 1024, generator=g) * 0.5
    mot = torch.randn(B, P, 1024, generator=g) * 0.5
    alpha = torch.softmax(torch.randn(B, L, 2 * P, generator=g), -1)
    eps = torch.rand(B, generator=g)
    return args, D, caps, logits_tm, obj, mot, alpha, eps


def oracle_grads(args, D, caps, logits_tm, obj, mot, alpha, eps):
    """loss_D of run_gun.py:345-375 and every critic gradient from the oracle's autograd in float64"""
    from oracle import gan_ref as GR
    V = logits_tm.shape[2]
    O = GR.DiscV2Ref(args, V).double().eval()
    O.load_state_dict({k: v.double() for k, v in D.state_dict().items()})
    f = logits_tm.transpose(0, 1).double()
    r = GR.to_onehot(caps, V).double()
    att = GR.attention_mask(caps).double()
    loss, r_loss, f_loss, gp, logit = GR.critic_losses(O, r, f, obj.double(), mot.double(), att, alpha.double(), eps.view(-1, 1, 1).double())
    loss.backward()
    return dict(loss=loss.item(), r=r_loss.item(), f=f_loss.item(), gp=gp.item(), out=torch.cat(logit).detach(),
                grads={k: p.grad for k, p in O.named_parameters()})


def check_grads(D, want, rel=2e-4):
    G = D.grad_views()
    for k, p in D.named_parameters():
        ref = want[k]
        ref = torch.zeros_like(p, dtype=torch.float64) if ref is None else ref
        got = G[k].double()
        err = (got - ref).abs().max().item()
        assert err <= rel * max(ref.abs().max().item(), 1e-3) + 1e-7, (k, err, ref.abs().max().item())


@pytest.mark.parametrize('topk,P', [(3, 8), (5, 5)])
def test_critic_update_matches_the_oracles_double_backward(topk, P):
    args, D, caps, logits_tm, obj, mot, alpha, eps = make_case(topk=topk, P=P)
    D.eval()
    want = oracle_grads(args, D, caps, logits_tm, obj, mot, alpha, eps)
    eng = D.engine
    B, L, V = caps.shape[0], caps.shape[1], logits_tm.shape[2]
    smask = (caps > 0).float()
    ws = eng.prepare(caps.device, B, L, V, smask, 4)
    eng.proposals(ws, obj, mot, alpha, smask)
    D.flatten_parameters_()
    D._gflat.fill_(float("nan"))                       # every entry must be WRITTEN by the schedule
    stats = eng.update_gradients(ws, caps, logits_tm, eps, 0)
    assert abs(float(stats[0]) - want['loss']) <= 2e-5 * max(1.0, abs(want['loss'])), (float(stats[0]), want['loss'])
    assert abs(float(stats[3]) - want['gp']) <= 2e-5 * max(1.0, want['gp'])
    assert abs(float(stats[1]) - want['r']) <= 1e-5 and abs(float(stats[2]) - want['f']) <= 1e-5
    assert abs(float(stats[4]) - (want['r'] - want['f'])) <= 1e-5
    assert (eng._bufs(ws)['outv'].double() - want['out']).abs().max().item() <= 1e-5
    check_grads(D, want['grads'])
    # a second update on the same workspace (buffers hold the previous update's values) gives the same gradients
    G = D.grad_views()
    keep = {k: G[k].clone() for k, _ in D.named_parameters()}
    eng.update_gradients(ws, caps, logits_tm, eps, 0)
    for k, t in keep.items():
        assert torch.equal(t, G[k]), k


# ---------------------------------------------------------------------------------------------- train mode: autograd over the blocks
def ref_scores(P, h, esel, v, smask, ng, seed, pd, p_sa, att_size):
    """the critic's forward as a differentiable composition (float64): h (n,L,C) -> scores (n,); masks as the engine keys them"""
    n, L, C = h.shape
    B = n // ng
    m = lambda site, rows, N, p: EC._m(seed, site, rows, N, p, 0, h)
    x0 = torch.relu(h)
    xp = F.pad(x0, (0, 0, 1, 1))
    taps = torch.stack([xp[:, :-2], xp[:, 1:-1], xp[:, 2:]], dim=3).reshape(n, L, 3 * C)
    x1 = x0 + 0.3 * (taps @ P['Wc'].t() + P['bc'])
    xin = x1 @ P['W_ih'].t() + P['b_ih'] + P['b_hh']
    hh = h.new_zeros(n, C); cc = h.new_zeros(n, C)
    hs = []
    for t in range(L):
        i, f, g, o = (xin[:, t] + hh @ P['W_hh'].t()).chunk(4, 1)
        cc = torch.sigmoid(f) * cc + torch.sigmoid(i) * torch.tanh(g)
        hh = torch.sigmoid(o) * torch.tanh(cc)
        hs.append(hh)
    Hs = torch.stack(hs, 1)
    y = EC.cln_f(Hs.reshape(n * L, C), P['ln_g'], P['ln_b'], False, 1e-5, None, m(CR.SITE_LSTM, n * L, C, pd)).view(n, L, C)
    K, Q, Vv = y @ P['WK'].t(), y @ P['WQ'].t(), y @ P['WV'].t()
    w, ctx = EC.sa_f(K, Q, Vv, EC._full_mask(smask, n), 1.0 / math.sqrt(att_size))
    out = ctx @ P['Wo'].t()
    words = EC.cln_f(out.reshape(n * L, C), P['an_g'], P['an_b'], True, 1e-5, m(CR.SITE_SA, n * L, C, p_sa), None).view(n, L, C)
    sm = smask.repeat(ng, 1).unsqueeze(2)
    s_, wg_ = [], []
    for k in range(2):
        apre = words @ P['Wa'][k].t() + P['ba'][k]
        a = EC.cln_f(apre.reshape(n * L, C), P['aa_g'][k], P['aa_b'][k], True, 1e-5, None, None).view(n, L, C)
        Pm, wgt, aggpre = EC.pattn_f(a, esel[k].repeat(ng, 1, 1), sm, 1.0 / math.sqrt(C))
        T = aggpre.shape[1]
        agg = EC.cln_f(aggpre.reshape(n * T, C), P['pn_g'][k], P['pn_b'][k], True, 1e-5, None, m(CR.SITE_PSL + k, n * T, C, pd)).view(n, T, C)
        s_.append(torch.tanh(agg @ P['Ws'][k].t() + P['bs'][k])); wg_.append(wgt)
    adj, u, sent, fus = EC.tsum_f(words, P['theta'].reshape(-1), P['ts_g'], P['ts_b'], P['fusion'], 1e-5, m(CR.SITE_TSUM, n, C, pd))
    pair, score, both, outv = EC.score_f(torch.stack(v), torch.stack(s_), torch.stack([x.reshape(-1) for x in P['wc']]),
                                         torch.cat([x.reshape(1) for x in P['bcl']]), torch.stack(wg_), fus, ng)
    return outv


def ref_update(D, caps, logits_tm, obj, mot, alpha, eps, seed):
    eng = D.engine
    p32, _ = eng._params()
    leaves = {}

    def dbl(t):
        if isinstance(t, list):
            return [dbl(x) for x in t]
        key = t.data_ptr()
        if key not in leaves:
            leaves[key] = t.detach().double().clone().requires_grad_(True)
        return leaves[key]
    P = {k: dbl(v) for k, v in p32.items() if k != 'Wkqv'}
    B, L = caps.shape
    T = D.num_top if D.num_psl > D.num_top else D.num_psl
    Pn = D.num_psl
    pd = 0.3 if D.training else 0.0
    smask = (caps > 0).double()
    psl = [obj.double().reshape(B * Pn, -1), mot.double().reshape(B * Pn, -1)]
    esel, v = [], []
    a = alpha.double() * smask.unsqueeze(2)
    for k in range(2):
        e = EC.cln_f(psl[k] @ P['We'][k].t() + P['be'][k], P['eg'][k], P['eb'][k], True, 1e-5, None, None).view(B, Pn, -1)
        if D.num_psl > D.num_top:
            sl = slice(0, Pn) if k == 0 else slice(alpha.shape[2] - Pn, alpha.shape[2])
            top = a[:, :, sl].sum(1).topk(T, dim=-1).indices
            e = e.gather(1, top.unsqueeze(-1).expand(B, T, e.shape[-1]))
        esel.append(e)
        v.append(torch.tanh(e @ P['Wv'][k].t() + P['bv'][k]))
    hf = logits_tm.double().transpose(0, 1) @ P['Wvoc'].t() + P['bvoc']
    hr = P['Wvoc'].t()[caps] + P['bvoc']
    e_ = eps.double().view(-1, 1, 1)
    hm = e_ * hr + (1 - e_) * hf
    h = torch.cat([hr, hf, hm], 0)
    out = ref_scores(P, h, esel, v, smask, 3, seed, pd, float(D.att.dropout) if D.training else 0.0, D.att.attention_size)
    g = torch.autograd.grad(out[2 * B:].sum(), hm, create_graph=True)[0]
    Gm = P['Wvoc'] @ P['Wvoc'].t()
    gn = torch.sqrt(((g @ Gm) * g).sum((1, 2)).clamp_min(1e-24))
    gp = ((gn - 1) ** 2).mean()
    loss = out[B:2 * B].mean() - out[:B].mean() + 10 * gp
    loss.backward()
    grads = {}
    for name, pr in D.named_parameters():
        leaf = leaves.get(pr.data_ptr())
        grads[name] = None if leaf is None or leaf.grad is None else leaf.grad.view(pr.shape)
    return loss.item(), gp.item(), out.detach(), grads


@pytest.mark.parametrize('train', [False, True])
def test_critic_update_matches_autograd_over_the_blocks(train):
    args, D, caps, logits_tm, obj, mot, alpha, eps = make_case(B=2, L=7, V=23, seed=11)
    D.train(train)
    seed = 0x1234567
    loss, gp, out, grads = ref_update(D, caps, logits_tm, obj, mot, alpha, eps, seed)
    eng = D.engine
    B, L, V = caps.shape[0], caps.shape[1], logits_tm.shape[2]
    smask = (caps > 0).float()
    ws = eng.prepare(caps.device, B, L, V, smask, 4)
    eng.proposals(ws, obj, mot, alpha, smask)
    D.flatten_parameters_()
    D._gflat.fill_(float("nan"))
    stats = eng.update_gradients(ws, caps, logits_tm, eps, seed)
    assert (eng._bufs(ws)['outv'].double() - out).abs().max().item() <= 2e-5
    assert abs(float(stats[0]) - loss) <= 3e-5 * max(1.0, abs(loss)) and abs(float(stats[3]) - gp) <= 3e-5 * max(1.0, gp)
    check_grads(D, grads, rel=3e-4)


def test_critic_forward_is_differentiable_once_through_autograd():
    """DiscV2.forward(inputs, obj, mot, att_mask, alpha_all) (models/model.py:143) under torch autograd: scores, d scores / d inputs
    and every parameter gradient against the oracle; a double backward raises instead of returning a wrong number"""
    from oracle import gan_ref as GR
    args, D, caps, logits_tm, obj, mot, alpha, eps = make_case(seed=9)
    D.eval()
    V = logits_tm.shape[2]
    tokens = logits_tm.transpose(0, 1).contiguous().requires_grad_(True)
    att = GR.attention_mask(caps)
    out = D(tokens, obj, mot, att_mask=att, alpha_all=alpha)
    (-out.mean()).backward()
    O = GR.DiscV2Ref(args, V).double().eval()
    O.load_state_dict({k: v.double() for k, v in D.state_dict().items()})
    t64 = tokens.detach().double().requires_grad_(True)
    want = O(t64, obj.double(), mot.double(), att.double(), alpha.double())
    (-want.mean()).backward()
    assert (out.detach().double() - want.detach()).abs().max().item() <= 1e-5
    assert (tokens.grad.double() - t64.grad).abs().max().item() <= 2e-4 * t64.grad.abs().max().item() + 1e-8
    for (k, p), (_, q) in zip(D.named_parameters(), O.named_parameters()):
        ref = q.grad if q.grad is not None else torch.zeros_like(q)
        err = (p.grad.double() - ref).abs().max().item()
        assert err <= 2e-4 * max(ref.abs().max().item(), 1e-3) + 1e-7, (k, err)
    tokens2 = tokens.detach().clone().requires_grad_(True)
    out2 = D(tokens2, obj, mot, att_mask=att, alpha_all=alpha)
    with pytest.raises(RuntimeError):
        g = torch.autograd.grad(out2.sum(), tokens2, create_graph=True)[0]
        g.sum().backward()
