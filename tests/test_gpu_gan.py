"""GPU (-m gpu): SURVEY.md 8(f) rank 1 -- one RunGAN iteration (run_gun.py:153-234,339-398) with the generator AND the DiscV2
critic on the HIP kernels (dlsg_amd/gan.py, dlsg_amd/critic.py), against the reference's own numbers
(tests/golden/gan_*.npz: the imported reference CapGnnModel + DiscV2, five critic updates with recorded gradient-penalty
epsilons, then the generator step with total_loss = cap_loss + lambda * loss_G)."""
import numpy as np
import pytest
import torch

import dlsg_amd
from dlsg_amd import gan
from helpers import load_gan_case, check_post

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('tag', ['gan_msvd', 'gan_msrvtt'])
def test_gan_iteration_matches_reference(tag):
    """the critic's first update (scores, penalty, every gradient norm), five updates and the generator step of one RunGAN
    iteration, every launch one of this repo's kernels (critic.CriticEngine + Trainer)"""
    args, vocab, g, G, D, frames, regions, caps, lens = load_gan_case(tag, dlsg_amd.CapGnnModel, dlsg_amd.DiscV2)
    G, D = G.cuda(), D.cuda()
    frames, regions, caps = frames.cuda(), regions.cuda(), caps.cuda()
    eps = torch.from_numpy(g['eps_gp']).cuda()
    with torch.no_grad():
        f_caption, obj, mot, alpha = G(frames, regions, caps, 26, 1.0)
    import copy
    D0 = copy.deepcopy(D)
    B, L, V = caps.shape[0], caps.shape[1], int(g['meta.V'])
    smask = (caps > 0).float()
    eng = D0.engine
    ws = eng.prepare(caps.device, B, L, V, smask, 4)
    eng.proposals(ws, obj, mot, alpha, smask)
    stats = eng.update_gradients(ws, caps, f_caption.transpose(0, 1).contiguous(), eps[0].reshape(B), 0).cpu()
    outv = eng._bufs(ws)['outv'].cpu().numpy()
    assert np.abs(outv[:B] - g['d0.r_logit']).max() <= 2e-4
    assert np.abs(outv[B:2 * B] - g['d0.f_logit']).max() <= 2e-4
    assert np.abs(outv[2 * B:] - g['d0.mixed_logit']).max() <= 2e-4
    assert abs(float(stats[3]) - float(g['d0.gp'])) <= 1e-3 * max(1.0, float(g['d0.gp']))
    assert abs(float(stats[0]) - float(g['d0.loss_D'])) <= 2e-3
    Gd = D0.grad_views()
    for n, p in D0.named_parameters():
        ref = max(float(g['d0.gnorm.' + n]), 0.0)
        got = float(Gd[n].double().norm())
        assert abs(got - ref) <= 3e-3 * max(abs(ref), 1e-3), (n, got, ref)
    # the reference's call signature, dense one-hot input, through the autograd bridge
    with torch.no_grad():
        dense = D(torch.nn.functional.one_hot(caps, V).float(), obj, mot, gan.attention_mask(caps), alpha)
    assert np.abs(dense.cpu().numpy() - g['d0.r_logit']).max() <= 2e-4
    it = dlsg_amd.GanTrainer(G, D, num_D=int(g['meta.num_D']), gan_lambda=float(g['meta.lambda']))
    it.eps_source = lambda k: eps[k]
    res = it.iteration(frames, regions, caps, lens, 1.0)
    assert abs(res['loss_D'] - float(g['loss_D_mean'])) <= 3e-3
    assert abs(res['wasserstein'] - float(g['wasserstein_mean'])) <= 3e-3
    check_post(D.named_parameters(), g, 'dpost.', 2e-4)
    assert abs(res['cap_loss'] - float(g['cap_loss'])) <= 1e-4
    assert abs(res['loss_G'] - float(g['loss_G'])) <= 2e-3
    assert abs(res['total_loss'] - float(g['total_loss'])) <= 2e-4
    Gv = G.grad_views()
    for k, p in G.named_parameters():
        if 'gnorm.' + k in g:
            ref = float(g['gnorm.' + k])
            got = float(Gv[k].double().norm())
            assert abs(got - ref) <= 5e-3 * max(ref, 1e-6) + 1e-6, (k, got, ref)
    check_post(G.named_parameters(), g, 'post.', 1e-4)
    G.ops.check_persistent()


def test_proposal_and_attention_gradients_reach_the_encoder():
    """`obj_proposals`, `motion_proposals` and `alpha_all` are graph-attached outputs of forward() (models/model.py:36-40):
    a caller that does NOT detach them (unlike run_gun.py:172-174) sends d(obj), d(mot), d(alpha) back through the
    hand-scheduled backward.  Checked against the oracle's autograd with a loss that touches all four outputs."""
    from oracle import torch_ref as R
    from helpers import load_case, weights_and_inputs
    args, vocab, g, kind = load_case('small_msvd')
    torch.manual_seed(0)
    net = dlsg_amd.CapGnnModel(args, vocab).eval()
    sd, frames, regions, caps, lens = weights_and_inputs(net, g, args)
    net.load_state_dict(sd)
    orc = R.CapGnnModelRef(args, vocab).eval()
    orc.load_state_dict(sd)
    gen = torch.Generator().manual_seed(3)
    w_obj, w_mot = torch.randn(3, 8, 64, generator=gen), torch.randn(3, 8, 64, generator=gen)
    w_al, w_lg = torch.randn(3, 26, 16, generator=gen), torch.randn(3, 26, 50, generator=gen) * 0.1

    def loss_of(m, dev):
        out, obj, mot, alpha = m(frames.to(dev), regions.to(dev), caps.to(dev), 26, 1.0)
        return (out * w_lg.to(dev)).sum() + (obj * w_obj.to(dev)).sum() + (mot * w_mot.to(dev)).sum() + (alpha * w_al.to(dev)).sum()
    lo = loss_of(orc, 'cpu')
    lo.backward()
    net = net.cuda()
    lh = loss_of(net, 'cuda')
    lh.backward()
    assert abs(lo.item() - lh.item()) <= 1e-3 * max(1.0, abs(lo.item()))
    want = dict(orc.named_parameters())
    for k, p in net.named_parameters():
        if want[k].grad is None:
            continue
        ref = want[k].grad
        err = (p.grad.cpu() - ref).abs().max().item()
        assert err <= 2e-5 + 2e-3 * ref.abs().max().item(), (k, err, ref.abs().max().item())


def test_graph_replayed_critic_updates_match_eager_updates():
    """GanTrainer replays a critic update from hipGraphs (update_gradients | Adam); the replayed updates must equal the
    kernel-by-kernel ones -- same losses, same critic weights, bit for bit -- also across a batch of another shape in between."""
    import copy
    args, vocab, g, G, D, frames, regions, caps, lens = load_gan_case('gan_msvd', dlsg_amd.CapGnnModel, dlsg_amd.DiscV2)
    G, D = G.cuda(), D.cuda().train()
    frames, regions, caps = frames.cuda(), regions.cuda(), caps.cuda()
    eps = torch.rand(6, caps.shape[0], generator=torch.Generator().manual_seed(3)).cuda()
    with torch.no_grad():
        f_caption, obj, mot, alpha = G(frames, regions, caps, 26, 1.0)
    logits_tm = f_caption.transpose(0, 1).contiguous()
    smask = (caps > 0).float()
    runs = []
    for graphs in (False, True):
        Dk = copy.deepcopy(D)
        it = dlsg_amd.GanTrainer(G, Dk, num_D=2, use_graphs=graphs)
        log = []
        for call in range(4):
            it.eps_source = lambda k, call=call: eps[(2 * call + k) % 6]
            if call == 2:                                    # another batch shape
                it.eps_source = lambda k: eps[k][:2]
                log.append(it.train_disc(caps[:2], logits_tm[:, :2].contiguous(), obj[:2], mot[:2], smask[:2], alpha[:2]))
            else:
                log.append(it.train_disc(caps, logits_tm, obj, mot, smask, alpha))
        assert bool(it._cg) == graphs
        runs.append((log, [p.detach().clone() for p in Dk.parameters()]))
    (la, pa), (lb, pb) = runs
    assert la == lb, (la, lb)
    for a, b in zip(pa, pb):
        assert torch.equal(a, b)


def test_checkpoint_reload_drops_the_captured_critic_graphs(tmp_path):
    """Loading a checkpoint re-reads the critic's Adam state: a trainer that reloads its own checkpoint must continue exactly
    like a fresh trainer that loads it (captured graphs are dropped with the state they were captured on)."""
    import copy
    args, vocab, g, G, D, frames, regions, caps, lens = load_gan_case('gan_msvd', dlsg_amd.CapGnnModel, dlsg_amd.DiscV2)
    G, D = G.cuda(), D.cuda().eval()
    frames, regions, caps = frames.cuda(), regions.cuda(), caps.cuda()
    eps = torch.rand(2, caps.shape[0], generator=torch.Generator().manual_seed(5)).cuda()
    a = dlsg_amd.GanTrainer(G, D, num_D=2)
    a.eps_source = lambda k: eps[k]
    import random
    random.seed(1)
    for i in range(3):
        a.iteration(frames, regions, caps, lens, 1.0, 0, i + 1)
    assert a._cg                                                        # critic updates are replayed by now
    path = str(tmp_path / 'ck.pt')
    dlsg_amd.save_checkpoint(path, 0, a)
    G2 = dlsg_amd.CapGnnModel(args, vocab).cuda().train(G.training)
    D2 = dlsg_amd.DiscV2(args, len(vocab)).cuda().eval()
    b = dlsg_amd.GanTrainer(G2, D2, num_D=2)
    b.eps_source = a.eps_source
    G2.seed_counter = G.seed_counter
    dlsg_amd.load_checkpoint(path, b)
    dlsg_amd.load_checkpoint(path, a)
    assert not a._cg
    outs = []
    for t in (a, b):
        random.seed(2)
        outs.append([t.iteration(frames, regions, caps, lens, 1.0, 0, 4 + i) for i in range(2)])
    for x, y in zip(*outs):
        for k in ('cap_loss', 'loss_G', 'loss_D', 'wasserstein'):
            assert abs(x[k] - y[k]) <= 2e-5 * max(1.0, abs(x[k])), (k, x, y)
    for (k, p), (_, q) in zip(D.named_parameters(), D2.named_parameters()):
        d = (p - q).abs()
        assert d.max().item() <= 4 * 2 * 1.6e-4 and (d > 2e-6).float().mean().item() <= 5e-3, k


def test_replayed_generator_forward_equals_the_model_call():
    """GanTrainer takes the no-grad generator forward of run_gun.py:167 from the captured step's first graph once it exists:
    with the same coin / seed draws it must return what `model(frames, regions, captions, 26, tf)` returns."""
    import random
    args, vocab, g, G, D, frames, regions, caps, lens = load_gan_case('gan_msvd', dlsg_amd.CapGnnModel, dlsg_amd.DiscV2)
    G, D = G.cuda().train(), D.cuda()
    frames, regions, caps = frames.cuda(), regions.cuda(), caps.cuda()
    it = dlsg_amd.GanTrainer(G, D, num_D=1)
    random.seed(4)
    assert it.trainer.forward_only(frames, regions, caps, 0.7) is None          # nothing captured yet
    it.iteration(frames, regions, caps, lens, 0.7, 0, 1)
    state, counter = random.getstate(), G.seed_counter
    got = [t.clone() for t in it.trainer.forward_only(frames, regions, caps, 0.7)]
    assert G.seed_counter == counter + 1
    random.setstate(state)
    G.seed_counter = counter
    with torch.no_grad():
        want = G(frames, regions, caps, 26, 0.7)
    for a, b in zip(got, want):
        assert a.shape == b.shape and torch.equal(a, b)
    assert it.trainer.forward_only(frames[:2], regions[:2], caps[:2], 0.7) is None       # another batch shape: the model call
