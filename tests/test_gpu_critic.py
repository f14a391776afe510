"""The critic-schedule kernels (csrc/critic_sched.hip, csrc/critic_lstm.hip) through the C ABI against their CPU emulation
(tests/emul_critic.py: explicit forward / backward formulas, third level by torch.func.jvp), block by block, and the whole
critic update (dlsg_amd/critic.py) on the GPU against the same schedule on the emulation and against torch autograd."""
import math

import numpy as np
import pytest
import torch

import dlsg_amd
from dlsg_amd.synth import synth_state_dict
from emul_ops import EmulOps
from helpers import gan_args

pytestmark = pytest.mark.gpu
C = 512


def dev(x):
    if torch.is_tensor(x):
        return x.cuda()
    if isinstance(x, (list, tuple)):
        return type(x)(dev(v) for v in x)
    return x


def flat(x, out):
    if torch.is_tensor(x):
        out.append(x)
    elif isinstance(x, (list, tuple)):
        for v in x:
            flat(v, out)
    return out


def both(name, args, kw=None, tol=2e-4, atol=2e-5, skip=()):
    """run ops.<name>(*args, **kw) on the emulation (CPU) and on the HIP library (copies on the device); compare every tensor"""
    kw = kw or {}
    cpu = [a for a in args]
    gpu = dev(args)
    kwg = {k: dev(v) for k, v in kw.items()}
    getattr(EmulOps(), name)(*cpu, **kw)
    getattr(hip_ops(), name)(*gpu, **kwg)
    torch.cuda.synchronize()
    a, b = flat(cpu, []) + flat(list(kw.values()), []), flat(gpu, []) + flat(list(kwg.values()), [])
    for i, (x, y) in enumerate(zip(a, b)):
        if i in skip:
            continue
        if x.dtype in (torch.int64, torch.int32):
            assert torch.equal(x, y.cpu()), (name, i)
            continue
        err = (x - y.cpu()).abs().max().item()
        assert err <= atol + tol * x.abs().max().item(), (name, i, err, x.abs().max().item())


_HIP = []


def hip_ops():
    if not _HIP:
        from dlsg_amd.hip import HipOps
        _HIP.append(HipOps())
    return _HIP[0]


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g) * scale


def smask_of(B, L, seed=0):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(2, L + 1, (B,), generator=g)
    lens[0] = L
    return (torch.arange(L).unsqueeze(0) < lens.unsqueeze(1)).float()


@pytest.mark.parametrize('B,L,V', [(3, 9, 37), (5, 26, 1000)])
def test_vocabulary_glue(B, L, V):
    W, bias = rnd(C, V, scale=0.1), rnd(C, scale=0.1)
    proj, eps = rnd(L, B, C), torch.rand(B)
    ids = torch.randint(0, V, (B, L))
    ids[:, L // 2:] = 0                                        # many rows share <pad>
    both('crit_embed_mix', [proj, ids, W, bias, eps, torch.zeros(3, B, L, C)])
    both('crit_embed_mix', [proj, None, W, bias, None, torch.zeros(1, B, L, C)])
    both('crit_embed_mix_bwd', [rnd(3, B, L, C), eps, torch.zeros(B, L, C), torch.zeros(L, B, C)])
    both('crit_embed_mix_bwd', [rnd(1, B, L, C), None, None, torch.zeros(L, B, C)])
    both('crit_vocab_scatter', [rnd(B, L, C), ids, rnd(C, V, seed=3)])


def test_vocab_scatter_is_bit_reproducible():
    B, L, V = 64, 26, 1000
    ids = torch.randint(0, V, (B, L))
    ids[:, 10:] = 0
    dhr = rnd(B, L, C).cuda()
    outs = []
    for _ in range(3):
        dW = torch.zeros(C, V, device='cuda')
        hip_ops().crit_vocab_scatter(dhr, ids.cuda(), dW)
        outs.append(dW.cpu())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    ref = torch.zeros(C, V)
    EmulOps().crit_vocab_scatter(dhr.cpu(), ids, ref)
    assert (ref - outs[0]).abs().max().item() <= 2e-4 * ref.abs().max().item()


@pytest.mark.parametrize('n,L', [(4, 9), (7, 26)])
def test_relu_taps(n, L):
    x, ref = rnd(n, L, C), rnd(n, L, C, seed=1)
    both('crit_relu_taps', [x, ref, rnd(C), 0.3, torch.zeros(n, L, C), torch.zeros(n, L, 3 * C)])
    both('crit_relu_taps', [x, ref, None, 0.0, torch.zeros(n, L, C), torch.zeros(n, L, 3 * C)])
    both('crit_relu_taps_bwd', [rnd(n, L, C, seed=2), rnd(n, L, 3 * C, seed=3), ref, torch.zeros(n, L, C)])


@pytest.mark.parametrize('n,L,H', [(5, 7, 64), (70, 26, 512), (192, 26, 512), (300, 5, 64)])
def test_lstm_sequence_levels(n, L, H):
    W = rnd(4 * H, H, scale=1.0 / math.sqrt(H))
    xin = rnd(n, L, 4 * H)
    b_ih, b_hh = rnd(4 * H, scale=0.1), rnd(4 * H, seed=1, scale=0.1)
    As, Hs, Cs, Hp = (torch.zeros(n, L, 4 * H), torch.zeros(n, L, H), torch.zeros(n, L, H), torch.zeros(n, L, H))
    both('lstm_seq_fwd', [xin, W, b_ih, b_hh, As, Hs, Cs, Hp], tol=3e-4)
    EmulOps().lstm_seq_fwd(xin, W, b_ih, b_hh, As, Hs, Cs, Hp)
    dHs, dAs, dCs = rnd(n, L, H, seed=2), rnd(n, L, 4 * H, seed=3, scale=0.1), rnd(n, L, H, seed=4, scale=0.1)
    DA, DH, DC = torch.zeros(n, L, 4 * H), torch.zeros(n, L, H), torch.zeros(n, L, H)
    both('lstm_seq_bwd', [As, Cs, W, dHs, dAs, dCs, DA, DH, DC], tol=5e-4)
    both('lstm_seq_bwd', [As, Cs, W, dHs, None, None, torch.zeros(n, L, 4 * H), torch.zeros(n, L, H), torch.zeros(n, L, H)], tol=5e-4)
    EmulOps().lstm_seq_bwd(As, Cs, W, dHs, dAs, dCs, DA, DH, DC)
    both('lstm_seq_bwd2', [As, Cs, W, DH, DC, rnd(n, L, 4 * H, seed=5), torch.zeros(n, L, 4 * H), torch.ones(n, L, H),
                           torch.zeros(n, L, H), torch.zeros(n, L, H), torch.zeros(n, L, H)], tol=1e-3)


@pytest.mark.parametrize('pre_tanh', [True, False])
@pytest.mark.parametrize('rows,N,G,drop', [(37, 512, 1, 0.0), (50, 512, 2, 0.3), (9, 1024, 1, 0.3), (4992, 512, 1, 0.3)])
def test_layernorm_levels(rows, N, G, drop, pre_tanh):
    x = [rnd(rows, N, seed=g) for g in range(G)]
    gamma = [1 + 0.1 * rnd(N, seed=10 + g) for g in range(G)]
    beta = [0.1 * rnd(N, seed=20 + g) for g in range(G)]
    kw = dict(p_pre=drop if pre_tanh else 0.0, site_pre=3, p_post=0.0 if pre_tanh else drop, site_post=5, seed=0x1234567, row0=17)
    both('cln_fwd', [x, gamma, beta, [torch.zeros(rows, N) for _ in range(G)], pre_tanh], kw)
    dys = [[rnd(rows, N, seed=30 + k + g) for g in range(G)] for k in range(3)]
    both('cln_bwd', [x, gamma, dys, [torch.zeros(rows, N) for _ in range(G)], [torch.zeros(N) for _ in range(G)],
                     [torch.zeros(N) for _ in range(G)], pre_tanh], dict(kw, extra=[rnd(2, N, seed=40 + g) for g in range(G)]),
         tol=5e-4, atol=1e-4)
    lo, hi = rows // 3, rows // 2 + 1
    both('cln_bwd', [x, gamma, dys[:1], [rnd(rows, N, seed=50 + g) for g in range(G)], None, None, pre_tanh], dict(kw, acc=(lo, hi)))
    both('cln_bwd2', [x, gamma, dys[:2], [rnd(rows, N, seed=60 + g) for g in range(G)], [torch.zeros(rows, N) for _ in range(G)],
                      [torch.zeros(rows, N) for _ in range(G)], [torch.zeros(2, N) for _ in range(G)], pre_tanh], kw, tol=1e-3, atol=2e-4)


@pytest.mark.parametrize('rows,N,G', [(50, 512, 2), (1664, 512, 1), (4992, 512, 2)])
def test_layernorm_deferred_partials_sum_to_the_folded_gradients(rows, N, G):
    """cln_bwd / cln_bwd2 with defer_ws leave per-workgroup partials (G, 2, ws rows, N): their row sums are the (dgamma, dbeta) /
    the second-order dgamma the folding launch writes (critic.CriticEngine folds them with its other column sums)"""
    ops = hip_ops()
    x = dev([rnd(rows, N, seed=g) for g in range(G)])
    gamma = dev([1 + 0.1 * rnd(N, seed=10 + g) for g in range(G)])
    dys = dev([[rnd(rows, N, seed=30 + k + g) for g in range(G)] for k in range(2)])
    U = dev([rnd(rows, N, seed=60 + g) for g in range(G)])
    kw = dict(p_pre=0.3, site_pre=3, seed=0x1234567, row0=17)
    z = lambda *s: torch.zeros(*s, device='cuda')
    dx0, dx1 = [z(rows, N) for _ in range(G)], [z(rows, N) for _ in range(G)]
    dg, db = [z(N) for _ in range(G)], [z(N) for _ in range(G)]
    ops.cln_bwd(x, gamma, dys, dx0, dg, db, True, **kw)
    wsb = torch.full((G, 2, ops.cln_ws_rows(rows, N), N), float('nan'), device='cuda')
    ops.cln_bwd(x, gamma, dys, dx1, None, None, True, defer_ws=wsb, **kw)
    gp = [z(2, N) for _ in range(G)]
    gx0, gy0, gx1, gy1 = ([z(rows, N) for _ in range(G)] for _ in range(4))
    ops.cln_bwd2(x, gamma, dys, U, gx0, gy0, gp, True, **kw)
    wst = torch.full((G, 2, ops.cln_ws_rows(rows, N), N), float('nan'), device='cuda')
    ops.cln_bwd2(x, gamma, dys, U, gx1, gy1, None, True, defer_ws=wst, **kw)
    torch.cuda.synchronize()
    for g in range(G):
        assert torch.equal(dx0[g], dx1[g]) and torch.equal(gx0[g], gx1[g]) and torch.equal(gy0[g], gy1[g])
        for got, want in ((wsb[g, 0].sum(0), dg[g]), (wsb[g, 1].sum(0), db[g]), (wst[g, 0].sum(0), gp[g][0])):
            assert torch.isfinite(got).all()
            assert (got - want).abs().max().item() <= 1e-4 + 1e-5 * want.abs().max().item()


@pytest.mark.parametrize('n,B,L', [(6, 3, 9), (8, 4, 26)])
def test_self_attention_levels(n, B, L):
    KQV = rnd(n, L, 3 * C, scale=0.7)
    sm = smask_of(B, L)
    sc = 1.0 / math.sqrt(C)
    w, ctx = torch.zeros(n, L, L), torch.zeros(n, L, C)
    both('crit_sa_fwd', [KQV, sm, w, ctx, sc])
    EmulOps().crit_sa_fwd(KQV, sm, w, ctx, sc)
    dctx = rnd(n, L, C, seed=1)
    both('crit_sa_bwd', [KQV, sm, w, dctx, torch.zeros(n, L, 3 * C), sc], tol=5e-4)
    both('crit_sa_bwd', [KQV, sm, w, dctx, rnd(n, L, 3 * C, seed=2), sc], dict(acc=(B, 2 * B)), tol=5e-4)
    both('crit_sa_bwd2', [KQV, sm, w, dctx, rnd(n, L, 3 * C, seed=3), torch.zeros(n, L, C), torch.zeros(n, L, 3 * C), sc], tol=1e-3, atol=1e-4)


@pytest.mark.parametrize('n,B,L,T', [(6, 3, 9, 3), (8, 4, 26, 5), (4, 4, 26, 8)])
def test_proposal_graph_levels(n, B, L, T):
    a = [rnd(n, L, C, seed=h) for h in range(2)]
    e = [rnd(B, T, C, seed=5 + h) for h in range(2)]
    sm = smask_of(B, L)
    sc = 1.0 / math.sqrt(C)
    z = lambda *s: [torch.zeros(*s) for _ in range(2)]
    P, wgt, agg = z(n, L, T), z(n, T), z(n, T, C)
    both('crit_pattn_fwd', [a, e, sm, P, wgt, agg, sc])
    EmulOps().crit_pattn_fwd(a, e, sm, P, wgt, agg, sc)
    d_agg, d_wgt = [rnd(n, T, C, seed=7 + h) for h in range(2)], [rnd(n, T, seed=9 + h) for h in range(2)]
    both('crit_pattn_bwd', [a, e, sm, P, d_agg, d_wgt, z(n, L, C), z(n, T, C), sc], tol=5e-4)
    both('crit_pattn_bwd', [a, e, sm, P, d_agg, d_wgt, [rnd(n, L, C, seed=11 + h) for h in range(2)], None, sc], dict(acc=(B, n)), tol=5e-4)
    Ua = [rnd(n, L, C, seed=13 + h) for h in range(2)]
    both('crit_pattn_bwd2', [a, e, sm, P, d_agg, d_wgt, Ua, z(n, T, C), z(n, T), z(n, L, C), z(n, T, C), sc], tol=1e-3, atol=1e-4)


@pytest.mark.parametrize('n,L,drop', [(5, 9, 0.0), (6, 26, 0.3)])
def test_text_summary_levels(n, L, drop):
    words = rnd(n, L, C)
    theta, gamma, beta, fusion = rnd(1, C, scale=0.1), 1 + 0.1 * rnd(C, seed=1), 0.1 * rnd(C, seed=2), rnd(2, C, seed=3, scale=0.1)
    kw = dict(p=drop, site=5, seed=99, row0=4)
    adj, u, sent, fus = torch.zeros(n, L), torch.zeros(n, C), torch.zeros(n, C), torch.zeros(n, 2)
    both('crit_tsum_fwd', [words, theta, gamma, beta, fusion, adj, u, sent, fus], kw)
    EmulOps().crit_tsum_fwd(words, theta, gamma, beta, fusion, adj, u, sent, fus, **kw)
    d_fus = rnd(n, 2, seed=4)
    both('crit_tsum_bwd', [words, theta, gamma, beta, fusion, adj, u, sent, fus, d_fus, torch.zeros(n, L, C), torch.zeros(n, 5, C)], kw, tol=5e-4)
    both('crit_tsum_bwd', [words, theta, gamma, beta, fusion, adj, u, sent, fus, d_fus, rnd(n, L, C, seed=5), None], dict(kw, acc=(1, 3)), tol=5e-4)
    both('crit_tsum_bwd2', [words, theta, gamma, beta, fusion, d_fus, rnd(n, L, C, seed=6), torch.zeros(n, 2), torch.zeros(n, L, C),
                            torch.zeros(n, 5, C)], kw, tol=1e-3, atol=1e-4)


@pytest.mark.parametrize('B,T,ng', [(3, 3, 3), (4, 5, 1), (64, 3, 3)])
def test_score_levels(B, T, ng):
    n = ng * B
    v = [torch.tanh(rnd(B, T, C, seed=h)) for h in range(2)]
    s = [torch.tanh(rnd(n, T, C, seed=2 + h)) for h in range(2)]
    wc, bc = [rnd(1, C, seed=4 + h, scale=0.1) for h in range(2)], [rnd(1, seed=6 + h) for h in range(2)]
    wgt = [torch.rand(n, T, generator=torch.Generator().manual_seed(8 + h)) + 0.2 for h in range(2)]
    fus = torch.softmax(rnd(n, 2, seed=10), 1)
    z = lambda *sh: [torch.zeros(*sh) for _ in range(2)]
    pair, score, bth, out = z(n, T), z(n), torch.zeros(ng, 2), torch.zeros(n)
    both('crit_score_fwd', [v, s, wc, bc, wgt, fus, pair, score, bth, out, ng])
    EmulOps().crit_score_fwd(v, s, wc, bc, wgt, fus, pair, score, bth, out, ng)
    d_out = rnd(n, seed=11)
    both('crit_score_bwd', [v, s, wc, wgt, fus, pair, score, bth, d_out, torch.zeros(n, 2), z(n, T, C), z(n, T, C), z(n, T), z(n, C),
                            torch.zeros(2), ng], tol=5e-4)
    both('crit_score_bwd', [v, s, wc, wgt, fus, pair, score, bth, d_out, rnd(n, 2, seed=12), [rnd(n, T, C, seed=13 + h) for h in range(2)],
                            None, [rnd(n, T, seed=15 + h) for h in range(2)], None, None, ng], dict(acc=(0, B)), tol=5e-4)
    if ng == 1:
        both('crit_score_bwd2', [v, s, wc, bc, wgt, fus, pair, score, d_out, [rnd(n, T, C, seed=17 + h) for h in range(2)],
                                 [rnd(n, T, seed=19 + h) for h in range(2)], rnd(n, 2, seed=21), torch.zeros(n, 2), z(n, T, C), z(n, T, C),
                                 z(n, T), z(n, C), torch.zeros(2)], tol=1e-3, atol=1e-4)


def test_penalty_topk_unselect_colsum():
    B, L = 5, 9
    g_, gG = rnd(B, L, C, scale=0.05), rnd(B, L, C, seed=1, scale=0.05)
    gG = gG.abs() * g_.sign()                                   # q_b > 0
    gG[1] = 0                                                   # one clamped sample
    both('crit_gp', [g_, gG, rnd(3 * B, seed=2), torch.zeros(8), torch.zeros(B, L, C), torch.zeros(B, L, C)])
    P, T = 8, 3
    alpha = torch.softmax(rnd(L, B, 2 * P, seed=3), -1).transpose(0, 1)        # strided, as the decoder's ALPHA arrives
    sm = smask_of(B, L)
    idx = torch.zeros(2, B, T, dtype=torch.int64)
    both('crit_topk', [alpha, sm, P, T, idx])
    EmulOps().crit_topk(alpha, sm, P, T, idx)
    both('crit_unselect', [rnd(2 * B * T, C, seed=4), idx.view(-1), torch.ones(2 * B * P, C), P])
    a, b2 = rnd(700, C, seed=5), rnd(40, C, seed=6)
    big = rnd(4992, 5, C, seed=7)
    one = rnd(2, seed=8)
    both('crit_colsum', [[([a], torch.zeros(C), torch.zeros(C), 1.0), ([a, b2], torch.zeros(C), None, 0.3),
                          ([big[:, 2]], torch.zeros(C), None, 1.0), ([one[1:2].view(1, 1), one[0:1].view(1, 1)], torch.zeros(1), None, 1.0)]],
         tol=5e-4, atol=1e-4)


# ---------------------------------------------------------------------------------------------- the whole update
def engine_case(B, L, V, seed, topk=3, P=8, train=False):
    args = gan_args(num_topk=topk, num_proposals=P)
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(2, L + 1, (B,), generator=g)
    lens[0] = L
    caps = torch.randint(4, V, (B, L), generator=g)
    for b in range(B):
        caps[b, int(lens[b]):] = 0
    logits_tm = torch.randn(L, B, V, generator=g) * 2
    obj, mot = torch.randn(B, P, 1024, generator=g) * 0.5, torch.randn(B, P, 1024, generator=g) * 0.5
    alpha = torch.softmax(torch.randn(B, L, 2 * P, generator=g), -1)
    eps = torch.rand(B, generator=g)
    Ds = []
    for where in ('cpu', 'cuda'):
        D = dlsg_amd.DiscV2(args, V)
        D.load_state_dict(synth_state_dict(D.state_dict(), seed + 1))
        D.train(train)
        if where == 'cpu':
            D.set_ops(EmulOps())
        Ds.append(D.to(where))
    return Ds, (caps, logits_tm, obj, mot, alpha, eps)


def run_update(D, data, seed):
    caps, logits_tm, obj, mot, alpha, eps = [t.to(next(D.parameters()).device) for t in data]
    eng = D.engine
    smask = (caps > 0).float()
    ws = eng.prepare(caps.device, caps.shape[0], caps.shape[1], logits_tm.shape[2], smask, 4)
    eng.proposals(ws, obj, mot, alpha, smask)
    D.flatten_parameters_()
    D._gflat.fill_(float('nan'))
    stats = eng.update_gradients(ws, caps, logits_tm, eps, seed)
    return stats.cpu(), {k: D.grad_views()[k].cpu() for k, _ in D.named_parameters()}, eng._bufs(ws)['outv'].cpu()


@pytest.mark.parametrize('B,L,V,topk,P,train', [(3, 9, 37, 3, 8, False), (4, 26, 60, 5, 5, True), (64, 26, 1000, 3, 8, True)])
def test_critic_update_on_the_gpu_matches_the_emulated_schedule(B, L, V, topk, P, train):
    (Dc, Dg), data = engine_case(B, L, V, 7, topk, P, train)
    sc, gc, oc = run_update(Dc, data, 0xABCDEF)
    sg, gg, og = run_update(Dg, data, 0xABCDEF)
    hip_ops().check_persistent()
    assert (oc - og).abs().max().item() <= 2e-4
    assert (sc[:5] - sg[:5]).abs().max().item() <= 3e-4 * max(1.0, sc[:5].abs().max().item())
    for k in gc:
        ref = gc[k]
        err = (ref - gg[k]).abs().max().item()
        assert err <= 2e-3 * max(ref.abs().max().item(), 1e-3) + 1e-6, (k, err, ref.abs().max().item())
    # bit-reproducible: the same update twice
    sg2, gg2, _ = run_update(Dg, data, 0xABCDEF)
    for k in gg:
        assert torch.equal(gg[k], gg2[k]), k


def test_slab_reduce_multi():
    a, b2 = rnd(8, 512, 1536, seed=1), rnd(4, 512, 37, seed=2)
    both('crit_reduce', [[(a, torch.zeros(512, 1536)), (b2, torch.zeros(512, 37)), (rnd(2, 512, 512, seed=3), torch.zeros(512, 512))]], tol=1e-5)
