"""CPU: the oracle (oracle/torch_ref.py) against golden vectors produced by the imported reference model.

This is what pins the oracle (tests/golden/make_goldens.py ran the reference itself).  Tolerance: 1e-5 abs on
logits (fp32, same torch CPU kernels, only reassociation from the K/V hoist), ids identical."""
import random

import numpy as np
import pytest
import torch

from oracle import torch_ref as R
from helpers import load_case, weights_and_inputs

SMALL = ['small_msvd', 'small_msrvtt', 'small_noobj', 'small_baseline1', 'small_baselinemodel']


def build(tag):
    args, vocab, g, kind = load_case(tag)
    torch.manual_seed(0)
    net = {'capgnn': R.CapGnnModelRef, 'baseline1': R.CapBaseline1Ref, 'baselinemodel': R.CapBaselineModelRef}[kind](args, vocab).eval()
    sd, frames, regions, caps, lens = weights_and_inputs(net, g, args)
    net.load_state_dict(sd, strict=True)
    return net, g, frames, regions, caps, lens, kind


@pytest.mark.parametrize('tag', SMALL)
def test_state_dict_keys_and_forward(tag):
    net, g, frames, regions, caps, lens, kind = build(tag)
    ref_keys = sorted(k[2:] for k in g if k.startswith('w.'))
    assert ref_keys == sorted(net.state_dict().keys())
    inter = {} if kind == 'capgnn' else None
    with torch.no_grad():
        if kind == 'capgnn':
            logits, obj, mot, alpha = net(frames, regions, caps, 26, 1.0, inter=inter)
        else:
            logits = net(frames, regions, caps, 26, 1.0)[0]
    assert np.abs(logits.numpy() - g['logits']).max() <= 1e-5
    if kind == 'capgnn':
        assert np.abs(obj.numpy() - g['obj_psl']).max() <= 1e-5
        assert np.abs(mot.numpy() - g['mot_psl']).max() <= 1e-5
        assert np.abs(alpha.numpy() - g['alpha']).max() <= 1e-5
        for k, v in inter.items():
            if 'i.' + k in g:
                assert np.abs(v.numpy() - g['i.' + k]).max() <= 1e-5, k


@pytest.mark.parametrize('tag', SMALL)
def test_greedy_and_beam_ids(tag):
    net, g, frames, regions, caps, lens, kind = build(tag)
    with torch.no_grad():
        net.update_beam_size(1)
        ids = net(frames, regions, None)[0]
        assert np.array_equal(ids.numpy(), g['greedy_ids'])
        net.update_beam_size(5)
        ids = net(frames, regions, None)[0]
        assert np.array_equal(ids.numpy(), g['beam5_ids'])


@pytest.mark.parametrize('tag', SMALL)
def test_scheduled_sampling_coin_order(tag):
    net, g, frames, regions, caps, lens, kind = build(tag)
    random.seed(12)
    with torch.no_grad():
        logits = net(frames, regions, caps, 26, 0.6)[0]
    assert np.abs(logits.numpy() - g['ss_logits']).max() <= 1e-5


@pytest.mark.parametrize('tag', SMALL)
def test_train_step_loss_grads_adam(tag):
    net, g, frames, regions, caps, lens, kind = build(tag)
    opt = R.make_optimizer(net)
    opt.zero_grad()
    outs = net(frames, regions, caps, 26, 1.0)[0]
    loss = R.ragged_ce(outs, caps, lens)
    loss.backward()
    assert abs(loss.item() - float(g['loss'])) <= 1e-5
    for k, p in net.named_parameters():
        if 'gnone.' + k in g:
            assert p.grad is None, k
        else:
            ref = g['g.' + k]
            assert np.abs(p.grad.numpy() - ref).max() <= 1e-5 + 1e-4 * np.abs(ref).max(), k
    opt.step()
    for k, p in net.named_parameters():
        s, a = g['post.' + k]
        assert abs(float(p.detach().double().sum()) - s) <= 1e-5 * max(1.0, a), k


def test_masked_self_attention(golden_dir):
    g = dict(np.load(golden_dir + '/sa_mask.npz'))
    m = R.SelfAttentionP(32, 32, 16, 0.3).eval()
    m.load_state_dict({k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('w.')})
    x = torch.from_numpy(g['x'])
    with torch.no_grad():
        y = R.self_attention(m, x, torch.from_numpy(g['mask']))
        y0 = R.self_attention(m, x)
    assert np.abs(y.numpy() - g['y_masked']).max() <= 1e-5
    assert np.abs(y0.numpy() - g['y']).max() <= 1e-5


@pytest.mark.parametrize('tag', ['full_msvd_b2'])
def test_full_size_msvd(tag):
    net, g, frames, regions, caps, lens, kind = build(tag)
    with torch.no_grad():
        logits = net(frames, regions, caps, 26, 1.0)[0]
        assert np.abs(logits.numpy() - g['logits']).max() <= 2e-5
        net.update_beam_size(1)
        assert np.array_equal(net(frames, regions, None)[0].numpy(), g['greedy_ids'])
        net.update_beam_size(5)
        assert np.array_equal(net(frames, regions, None)[0].numpy(), g['beam5_ids'])


def test_full_size_reference_default_feature_dims():
    """A = 1536, M = 1024 (utils/opt.py:69-70), tests/golden/make_goldens_r3.py: the oracle against the reference's top-8
    logits, ids, loss and gradient norms."""
    net, g, frames, regions, caps, lens, kind = build('full_default_b2')
    assert frames.shape[-1] == 2560
    with torch.no_grad():
        logits = net(frames, regions, caps, 26, 1.0)[0]
        top = torch.topk(logits, 8, dim=-1)
        assert np.array_equal(top.indices.numpy(), g['logits_top_idx'])
        assert np.abs(top.values.numpy() - g['logits_top_val']).max() <= 2e-5
        net.update_beam_size(1)
        assert np.array_equal(net(frames, regions, None)[0].numpy(), g['greedy_ids'])
        net.update_beam_size(5)
        assert np.array_equal(net(frames, regions, None)[0].numpy(), g['beam5_ids'])
    outs = net(frames, regions, caps, 26, 1.0)[0]
    loss = R.ragged_ce(outs, caps, lens)
    loss.backward()
    assert abs(loss.item() - float(g['loss'])) <= 1e-5
    for k, p in net.named_parameters():
        if 'gnorm.' + k in g:
            ref = float(g['gnorm.' + k])
            assert abs(float(p.grad.double().norm()) - ref) <= 1e-4 * max(ref, 1e-6) + 1e-7, k
        else:
            assert p.grad is None, k


@pytest.mark.parametrize('tag', ['small_msvd', 'small_msrvtt', 'small_baseline1', 'full_msvd_b2'])
def test_gradients_under_scheduled_sampling(tag):
    """tf = 0.6, random.seed(12): steps fed their own argmax (layer.py:432-439); loss and every gradient of the reference."""
    from helpers import load_aux, check_grads
    net, _, frames, regions, caps, lens, kind = build(tag)
    g = load_aux(tag, 'ss')
    random.seed(12)
    outs = net(frames, regions, caps, 26, 0.6)[0]
    loss = R.ragged_ce(outs, caps, lens)
    loss.backward()
    assert abs(loss.item() - float(g['loss'])) <= 1e-5
    if 'logits' in g:
        assert np.abs(outs.detach().numpy() - g['logits']).max() <= 1e-5
    check_grads(lambda k, p: p.grad, net.named_parameters(), g, rel=1e-4, abs_=1e-5)


@pytest.mark.parametrize('tag', ['gan_msvd', 'gan_msrvtt'])
def test_gan_critic_and_iteration(tag):
    """oracle/gan_ref.py against the reference's DiscV2 + one RunGAN iteration (tests/golden/gan_*.npz): critic scores,
    gradient penalty, the five critic steps, the generator step with cap_loss + lambda * loss_G."""
    import copy
    from oracle import gan_ref as GR
    from helpers import load_gan_case, check_post
    args, vocab, g, G, D, frames, regions, caps, lens = load_gan_case(tag, R.CapGnnModelRef, GR.DiscV2Ref)
    eps = torch.from_numpy(g['eps_gp'])
    V = int(g['meta.V'])
    with torch.no_grad():
        f_caption, obj, mot, alpha = G(frames, regions, caps, 26, 1.0)
    D0 = copy.deepcopy(D)
    loss_D, r_loss, f_loss, gp, (rl, fl, ml) = GR.critic_losses(D0, GR.to_onehot(caps, V), f_caption, obj, mot,
                                                                 GR.attention_mask(caps), alpha, eps[0])
    assert np.abs(rl.detach().numpy() - g['d0.r_logit']).max() <= 2e-5
    assert np.abs(fl.detach().numpy() - g['d0.f_logit']).max() <= 2e-5
    assert np.abs(ml.detach().numpy() - g['d0.mixed_logit']).max() <= 2e-5
    assert abs(gp.item() - float(g['d0.gp'])) <= 1e-4 * max(1.0, float(g['d0.gp']))
    assert abs(loss_D.item() - float(g['d0.loss_D'])) <= 2e-4
    loss_D.backward()
    for n, p in D0.named_parameters():
        ref = float(g['d0.gnorm.' + n])
        got = float(p.grad.double().norm()) if p.grad is not None else -1.0
        assert abs(got - ref) <= 2e-4 * max(abs(ref), 1e-3), (n, got, ref)
    opt_G, opt_D = R.make_optimizer(G), torch.optim.Adam(D.parameters(), lr=1.6e-4, betas=(0.5, 0.9))
    res = GR.gan_iteration(G, D, opt_G, opt_D, frames, regions, caps, lens, 1.0, float(g['meta.lambda']), int(g['meta.num_D']),
                           [eps[k] for k in range(eps.shape[0])])
    assert abs(res['loss_D'] - float(g['loss_D_mean'])) <= 5e-4
    assert abs(res['wasserstein'] - float(g['wasserstein_mean'])) <= 5e-4
    check_post(D.named_parameters(), g, 'dpost.', 2e-5)
    assert abs(res['cap_loss'] - float(g['cap_loss'])) <= 1e-5
    assert abs(res['loss_G'] - float(g['loss_G'])) <= 2e-4
    assert abs(res['total_loss'] - float(g['total_loss'])) <= 2e-5
    check_post(G.named_parameters(), g, 'post.', 1e-5)


def test_gan_lambda_handler_state_machine():
    """utils/utils.py:196-265 restated: stable until the running caption loss rises by 4 %, then one sine-shaped dip."""
    from oracle import gan_ref as GR
    h = GR.GANLambdaHandlerRef(total_step=100, gan_lambda=0.01)
    for i in range(1, 201):
        h.update_gan_lambda(0, i, 3.0)
        assert h.get_current_lambda() == 0.01
    for i in range(201, 330):
        h.update_gan_lambda(2, i - 200, 3.5)
    assert h.state == 1
    lam = [h.get_current_lambda() for _ in range(5)]
    assert lam[0] <= 0.01 and all(a >= b for a, b in zip(lam, lam[1:]))
    assert abs(min(h.decrease_schedule) - 0.006) < 1e-4 and abs(max(h.decrease_schedule) - 0.01) < 1e-4


def test_decoder_forward_with_step_feats():
    """models/layer.py:394,404-405 (`step_feats`): the oracle's decoder with a given global feature against the reference's
    (tests/golden/small_stepfeats.npz, make_goldens_r5.py)"""
    import os
    net, g, frames, regions, caps, lens, kind = build('small_msvd')
    fx = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'small_stepfeats.npz')))
    step = torch.from_numpy(fx['step_feats'])
    with torch.no_grad():
        obj, mot = R.capgnn_encoder(net.encoder, frames, regions)
        logits, _ = R.decoder_forward(net.decoder, obj, caps, 26, 1.0, feats2=mot, step_feats=step)
        assert np.abs(logits.numpy() - fx['logits']).max() <= 1e-5
        net.update_beam_size(1)
        ids, _ = R.decoder_forward(net.decoder, obj, None, 26, 1.0, feats2=mot, step_feats=step)
        assert np.array_equal(ids.numpy(), fx['greedy_ids'])


def test_decoder_forward_beam_branch_on_its_own():
    """models/layer.py:449-460: `Decoder.forward(cnn_feats, None, ...)` with beam_size != 1 runs the beam search itself -- the
    oracle's decoder against the reference's (tests/golden/small_decbeam.npz, make_goldens_r6.py), with and without `step_feats`"""
    import os
    net, g, frames, regions, caps, lens, kind = build('small_msvd')
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    fx = dict(np.load(os.path.join(here, 'small_decbeam.npz')))
    step = torch.from_numpy(np.load(os.path.join(here, 'small_stepfeats.npz'))['step_feats'])
    with torch.no_grad():
        obj, mot = R.capgnn_encoder(net.encoder, frames, regions)
        for k in (3, 5):
            net.update_beam_size(k)
            ids, _ = R.decoder_forward(net.decoder, obj, None, None, 1.0, feats2=mot)
            assert np.array_equal(ids.numpy(), fx['beam%d_ids' % k])
            ids, _ = R.decoder_forward(net.decoder, obj, None, None, 1.0, feats2=mot, step_feats=step)
            assert np.array_equal(ids.numpy(), fx['beam%d_ids_stepfeats' % k])
