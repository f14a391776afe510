"""CPU: host logic of the engine (launch schedule, buffer layout, hand-written backward) checked against the golden
vectors of the reference, with the kernels replaced by the test-only torch emulation (tests/emul_ops.py).
The same checks run against the real HIP kernels in tests/test_gpu_parity.py (-m gpu)."""
import random

import numpy as np
import pytest
import torch

import dlsg_amd
from emul_ops import EmulOps
import os
from helpers import load_case, weights_and_inputs, small_args

SMALL = ['small_msvd', 'small_msrvtt', 'small_noobj', 'small_baseline1', 'small_baselinemodel']
MODELS = {'capgnn': dlsg_amd.CapGnnModel, 'baseline1': dlsg_amd.CapBaseline1, 'baselinemodel': dlsg_amd.CapBaselineModel}


def build(tag, fused=True):
    args, vocab, g, kind = load_case(tag)
    torch.manual_seed(0)
    net = MODELS[kind](args, vocab).eval()
    net.set_ops(EmulOps(fused_supported=fused))
    sd, frames, regions, caps, lens = weights_and_inputs(net, g, args)
    net.load_state_dict(sd, strict=True)
    return net, g, frames, regions, caps, lens, kind


@pytest.mark.parametrize('tag', SMALL)
@pytest.mark.parametrize('fused', [True, False])
def test_forward_matches_reference(tag, fused):
    net, g, frames, regions, caps, lens, kind = build(tag, fused)
    assert sorted(net.state_dict().keys()) == sorted(k[2:] for k in g if k.startswith('w.'))
    with torch.no_grad():
        out = net(frames, regions, caps, 26, 1.0)
    assert np.abs(out[0].numpy() - g['logits']).max() <= 2e-5
    if kind == 'capgnn':
        assert np.abs(out[1].numpy() - g['obj_psl']).max() <= 2e-5
        assert np.abs(out[2].numpy() - g['mot_psl']).max() <= 2e-5
        assert np.abs(out[3].numpy() - g['alpha']).max() <= 2e-5


@pytest.mark.parametrize('tag', SMALL)
def test_greedy_ids_and_scheduled_sampling(tag):
    net, g, frames, regions, caps, lens, kind = build(tag)
    net.update_beam_size(1)
    with torch.no_grad():
        ids = net(frames, regions, None)[0]
    assert np.array_equal(ids.numpy(), g['greedy_ids'])
    random.seed(12)
    with torch.no_grad():
        logits = net(frames, regions, caps, 26, 0.6)[0]
    assert np.abs(logits.numpy() - g['ss_logits']).max() <= 2e-5


@pytest.mark.parametrize('tag', SMALL)
def test_autograd_path_grads(tag):
    net, g, frames, regions, caps, lens, kind = build(tag)
    outs = net(frames, regions, caps, 26, 1.0)[0]
    rows = torch.cat([outs[j][:lens[j]] for j in range(outs.shape[0])], 0)
    tgt = torch.cat([caps[j][:lens[j]] for j in range(outs.shape[0])], 0)
    loss = torch.nn.functional.cross_entropy(rows, tgt)
    loss.backward()
    assert abs(loss.item() - float(g['loss'])) <= 1e-5
    for k, p in net.named_parameters():
        if 'gnone.' + k in g:
            assert p.grad is None, k
        else:
            ref = g['g.' + k]
            assert np.abs(p.grad.numpy() - ref).max() <= 2e-5 + 2e-4 * np.abs(ref).max(), k


@pytest.mark.parametrize('tag', SMALL)
def test_trainer_step_matches_reference_adam(tag):
    net, g, frames, regions, caps, lens, kind = build(tag)
    tr = dlsg_amd.Trainer(net)
    loss = tr.step(frames, regions, caps, lens, 1.0)
    assert abs(float(loss) - float(g['loss'])) <= 1e-5
    for k, p in net.named_parameters():
        s, a = g['post.' + k]
        assert abs(float(p.detach().double().sum()) - s) <= 2e-5 * max(1.0, a), k


def test_train_mode_dropout_backward_is_consistent():
    """dropout on: the analytic gradient must match a directional finite difference of the same (seeded) loss."""
    net, g, frames, regions, caps, lens, kind = build('small_msvd')
    net.train()
    net = net.double() if False else net
    tr = dlsg_amd.Trainer(net, lr=0.0)
    net.seed_counter = 41
    random.seed(3)
    tr.step(frames, regions, caps, lens, 1.0)
    grad = net._gflat.clone()
    torch.manual_seed(5)
    d = torch.randn_like(grad) * (grad != 0)
    d /= d.norm()
    base = net._flat.clone()

    def loss_at(eps):
        net._flat.copy_(base + eps * d)
        net.seed_counter = 41
        random.seed(3)
        return float(tr.step(frames, regions, caps, lens, 1.0))
    h = 2e-2
    fd = (loss_at(h) - loss_at(-h)) / (2 * h)
    an = float((grad * d).sum())
    assert abs(fd - an) <= 0.05 * max(abs(an), 1e-3), (fd, an)


@pytest.mark.parametrize('tag', SMALL)
def test_beam5_ids(tag):
    net, g, frames, regions, caps, lens, kind = build(tag)
    net.update_beam_size(5)
    with torch.no_grad():
        ids = net(frames, regions, None)[0]
    assert np.array_equal(ids.numpy(), g['beam5_ids'])


@pytest.mark.parametrize('bias', [4.0, 7.0, 30.0])
@pytest.mark.parametrize('k', [5, 3])
def test_beam_early_exit_matches_oracle(bias, k):
    """<end> made likely: beams finish at different steps (the finished-beam rule, allennlp_beamsearch.py:147-150,186-190)
    and the search stops early (:168) -- the output is shorter than max_words; compared with the oracle's beam search."""
    from oracle import torch_ref as R
    net, g, frames, regions, caps, lens, kind = build('small_msvd')
    sd = {kk: v.clone() for kk, v in net.state_dict().items()}
    sd['decoder.word_restore.bias'][net.decoder.vocab('<end>')] += bias
    net.load_state_dict(sd)
    args, vocab, _, _ = load_case('small_msvd')
    orc = R.CapGnnModelRef(args, vocab).eval()
    orc.load_state_dict(sd)
    orc.update_beam_size(k)
    net.update_beam_size(k)
    with torch.no_grad():
        want = orc(frames, regions, None)[0]
        got = net(frames, regions, None)[0]
    assert got.shape == want.shape and torch.equal(got, want), (got.shape, want.shape)
    if bias >= 30.0:
        assert got.shape[1] < 26


@pytest.mark.parametrize('tag', ['small_msvd', 'small_baseline1'])
def test_device_coin_path_matches_host_coin_path(tag):
    """select_embed (coins applied on device, graph-invariant launch sequence) == the host-branching path."""
    outs = []
    for dev in (False, True):
        net, g, frames, regions, caps, lens, kind = build(tag)
        tr = dlsg_amd.Trainer(net, device_coins=dev)
        random.seed(12)
        loss = tr.step(frames, regions, caps, lens, 0.6)
        outs.append((float(loss), net._flat.clone()))
    assert abs(outs[0][0] - outs[1][0]) <= 1e-6
    # Adam normalises: elements whose gradient is rounding noise may move by a fraction of lr (1.6e-4)
    assert (outs[0][1] - outs[1][1]).abs().max().item() <= 3e-5


def test_optimizer_state_round_trips_with_torch_adam():
    """Checkpoint compatibility (run_gun.py:302-310 stores optimizer.state_dict()): the trainer's Adam state equals
    torch.optim.Adam's after the same steps on the oracle, exports in torch's layout, and a trainer resumed from
    torch's state continues identically."""
    from oracle import torch_ref as R
    net, g, frames, regions, caps, lens, kind = build('small_msvd')
    args, vocab, _, _ = load_case('small_msvd')
    orc = R.CapGnnModelRef(args, vocab).eval()
    orc.load_state_dict({k: v.clone() for k, v in net.state_dict().items()})
    opt = R.make_optimizer(orc)
    tr = dlsg_amd.Trainer(net)
    for _ in range(2):
        tr.step(frames, regions, caps, lens, 1.0)
        R.train_step(orc, opt, frames, regions, caps, lens, 1.0)
    mine, ref = tr.optimizer_state_dict(), opt.state_dict()
    assert sorted(mine['state'].keys()) == sorted(ref['state'].keys())
    for i, st in ref['state'].items():
        assert float(mine['state'][i]['step']) == float(st['step'])
        for key in ('exp_avg', 'exp_avg_sq'):
            scale = st[key].abs().max().item() + 1e-12
            assert (mine['state'][i][key] - st[key]).abs().max().item() <= 2e-4 * scale + 1e-10, (i, key)
    # torch accepts the exported layout
    opt2 = R.make_optimizer(orc)
    opt2.load_state_dict(mine)
    # resume a fresh trainer from torch's state: third step equals the oracle's third step
    net2, *_ = build('small_msvd')
    net2.load_state_dict({k: v.clone() for k, v in orc.state_dict().items()})
    tr2 = dlsg_amd.Trainer(net2)
    tr2.load_optimizer_state_dict(ref)
    assert tr2.t == 2
    tr2.step(frames, regions, caps, lens, 1.0)
    R.train_step(orc, opt, frames, regions, caps, lens, 1.0)
    for k, v in orc.state_dict().items():
        assert (net2.state_dict()[k] - v).abs().max().item() <= 2e-6, k


def test_caller_side_schedules_match_the_reference():
    """run_gun.py:94-95 (MultiStepLR([4,7], 0.5)) and :136 (scheduled-sampling epsilon)."""
    import math
    p = [torch.nn.Parameter(torch.zeros(1))]
    opt = torch.optim.Adam(p, lr=1.6e-4)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, [4, 7], 0.5)
    for epoch in range(10):
        assert abs(opt.param_groups[0]['lr'] - dlsg_amd.multistep_lr(epoch)) < 1e-12
        assert abs(dlsg_amd.ss_epsilon(epoch) - max(0.6, 20 / (20 + math.exp(epoch / 20)))) < 1e-12
        opt.step()
        sch.step()


@pytest.mark.parametrize('tag', ['small_msvd', 'small_msrvtt', 'small_baseline1'])
@pytest.mark.parametrize('dev_coins', [False, True])
def test_gradients_under_scheduled_sampling(tag, dev_coins):
    """tf = 0.6 with the reference's coin order: host-branching schedule and the device-coin schedule (select_embed /
    skip_if, what a replayed hipGraph runs) against the reference's loss and gradients (tests/golden/*_ss.npz)."""
    from helpers import load_aux, check_grads
    net, _, frames, regions, caps, lens, kind = build(tag)
    g = load_aux(tag, 'ss')
    tr = dlsg_amd.Trainer(net, lr=0.0, device_coins=dev_coins)
    random.seed(12)
    loss = tr.step(frames, regions, caps, lens, 0.6)
    assert abs(float(loss) - float(g['loss'])) <= 1e-5
    G = net.grad_views()
    check_grads(lambda k, p: G[k], net.named_parameters(), g, rel=2e-4)


@pytest.mark.parametrize('tag', ['small_msvd', 'small_msrvtt'])
@pytest.mark.parametrize('dev_coins', [False, True])
def test_dropout_placement_matches_reference(tag, dev_coins):
    """TRAIN mode.  tests/golden/*_drop.npz: the reference's own forward/backward with torch.nn.functional.dropout taking
    its masks from the build's counter hash, keyed by call order -> site.  A dropout at the wrong place, on the wrong
    tensor or with the wrong p in engine.py changes logits and gradients."""
    from helpers import load_aux, check_grads
    net, _, frames, regions, caps, lens, kind = build(tag)
    g = load_aux(tag, 'drop')
    net.train()
    net.seed_counter = int(g['meta.counter']) - 1
    tr = dlsg_amd.Trainer(net, lr=0.0, device_coins=dev_coins)
    random.seed(4)
    loss = tr.step(frames, regions, caps, lens, float(g['meta.tf']))
    assert abs(float(loss) - float(g['loss'])) <= 1e-5
    G = net.grad_views()
    check_grads(lambda k, p: G[k], net.named_parameters(), g, rel=2e-4)
    net.seed_counter = int(g['meta.counter']) - 1
    random.seed(4)
    with torch.no_grad():
        logits = net(frames, regions, caps, 26, float(g['meta.tf']))[0]
    assert np.abs(logits.numpy() - g['logits']).max() <= 2e-5


def test_load_encoder_grafts_and_freezes_word_embedding(tmp_path):
    """models/model.py:45-53 + one step: the grafted encoder / word embedding are the donor's, the frozen embedding stays
    bit-unchanged (Trainer skips requires_grad == False ranges in Adam and in the buckets), everything else equals the
    oracle doing the same with torch.optim.Adam."""
    from helpers import graft_and_step, OracleTrainer
    from oracle import torch_ref as R

    def mk(args, vocab):
        m = dlsg_amd.CapGnnModel(args, vocab)
        m.set_ops(EmulOps())
        return m
    net, emb0, loss = graft_and_step(mk, 'cpu', tmp_path, lambda m: dlsg_amd.Trainer(m))
    orc, emb0_o, loss_o = graft_and_step(R.CapGnnModelRef, 'cpu', tmp_path, OracleTrainer)
    assert abs(loss - loss_o) <= 1e-5
    assert torch.equal(net.decoder.word_embed.weight.detach(), emb0)
    assert not net.decoder.word_embed.weight.requires_grad
    want = dict(orc.named_parameters())
    moved = 0
    for k, p in net.named_parameters():
        d = (p.detach() - want[k].detach()).abs()
        # Adam's first step is lr * g / (|g| + eps): elements whose gradient is at rounding level (|g| ~ eps) may differ by up
        # to lr between two correct implementations; they are rare
        assert d.max().item() <= 3.3e-4 and (d > 2e-6).float().mean().item() <= 2e-3, (k, d.max().item())
        moved += int(not torch.equal(p.detach(), emb0) and p.requires_grad)
    assert moved > 50


def test_trainer_rebinds_when_a_parameter_is_frozen_later():
    net, g, frames, regions, caps, lens, kind = build('small_msvd')
    tr = dlsg_amd.Trainer(net)
    tr.step(frames, regions, caps, lens, 1.0)
    w = net.encoder.obj_encoder.obj_embed.weight
    w.requires_grad = False
    before = w.detach().clone()
    other = net.decoder.word_restore.weight.detach().clone()
    tr.step(frames, regions, caps, lens, 1.0)
    assert torch.equal(w.detach(), before)
    assert not torch.equal(net.decoder.word_restore.weight.detach(), other)
    sd = tr.optimizer_state_dict()
    names = [n for n, _ in net.named_parameters()]
    assert names.index('encoder.obj_encoder.obj_embed.weight') not in sd['state']


def test_model_deepcopies_and_pickles(tmp_path):
    import copy
    net, g, frames, regions, caps, lens, kind = build('small_msvd')
    with torch.no_grad():
        want = net(frames, regions, caps, 26, 1.0)[0]
    twin = copy.deepcopy(net)
    twin.set_ops(EmulOps())
    torch.save(net, str(tmp_path / 'm.pt'))
    back = torch.load(str(tmp_path / 'm.pt'), weights_only=False)
    back.set_ops(EmulOps())
    with torch.no_grad():
        assert torch.equal(twin(frames, regions, caps, 26, 1.0)[0], want)
        assert torch.equal(back(frames, regions, caps, 26, 1.0)[0], want)


def test_kernel_limits_are_reported_at_construction():
    from helpers import small_args
    with pytest.raises(ValueError, match='num_proposals'):
        dlsg_amd.CapGnnModel(small_args(num_proposals=80), dlsg_amd.make_vocab(50))
    with pytest.raises(ValueError, match='max_frames'):
        dlsg_amd.CapBaseline1(small_args(max_frames=73), dlsg_amd.make_vocab(50))
    dlsg_amd.CapBaseline1(small_args(max_frames=72), dlsg_amd.make_vocab(50))      # the reference's longest legal clip (sublayer.py:87)
    with pytest.raises(ValueError, match='decode_hidden_size'):
        dlsg_amd.CapGnnModel(small_args(decode_hidden_size=4096), dlsg_amd.make_vocab(50))


def test_deep_weight_gradient_split_respects_group_limit():
    """rows = 8450 (5 objects x 26 frames x 65 clips): the row split of gemm_tn_deep must stay within 16 groups per launch."""
    from dlsg_amd import engine as E
    ops = EmulOps()
    launches = []
    real = ops.gemm

    def spy(mode, groups, **kw):
        launches.append(len(groups))
        return real(mode, groups, **kw)
    ops.gemm = spy
    g = torch.Generator().manual_seed(0)
    items = []
    for _ in range(2):
        dy, x = torch.randn(8450, 8, generator=g), torch.randn(8450, 6, generator=g)
        items.append((dy, x, torch.zeros(8, 6)))
    E.gemm_tn_deep(ops, items, items[0][0])
    assert max(launches) <= 16
    for dy, x, gout in items:
        assert (gout - dy.t() @ x).abs().max().item() <= 1e-3


# ------------------------------------------------------------------ DiscV2 critic + WGAN-GP iteration
@pytest.mark.parametrize('tag', ['gan_msvd', 'gan_msrvtt'])
def test_gan_iteration_matches_reference(tag):
    """dlsg_amd.gan (embedding-gather projection, one 3B-row critic pass, Gram-matrix gradient penalty, unrolled LSTM, GAN
    term entering the hand-scheduled backward through d(logits)) against the reference's own numbers for one RunGAN
    iteration: critic scores, gradient penalty, five critic updates, generator loss / gradients / Adam step."""
    import copy
    from dlsg_amd import gan
    from helpers import load_gan_case, check_post

    def mk(args, vocab):
        m = dlsg_amd.CapGnnModel(args, vocab)
        m.set_ops(EmulOps())
        return m
    args, vocab, g, G, D, frames, regions, caps, lens = load_gan_case(tag, mk, lambda a, v: dlsg_amd.DiscV2(a, v).set_ops(EmulOps()))
    assert sorted(D.state_dict().keys()) == sorted([k[len('dpost.'):] for k in g if k.startswith('dpost.')] + ['att.pe.pe'])
    eps = torch.from_numpy(g['eps_gp'])
    with torch.no_grad():
        f_caption, obj, mot, alpha = G(frames, regions, caps, 26, 1.0)
    # one critic update's losses and gradients on a copy (critic.CriticEngine: F, B1, T, B2), against the reference's first update
    D0 = copy.deepcopy(D).set_ops(EmulOps())
    B, L, V = caps.shape[0], caps.shape[1], int(g['meta.V'])
    smask = (caps > 0).float()
    eng = D0.engine
    ws = eng.prepare(caps.device, B, L, V, smask, 4)
    eng.proposals(ws, obj, mot, alpha, smask)
    stats = eng.update_gradients(ws, caps, f_caption.transpose(0, 1).contiguous(), eps[0].reshape(B), 0)
    outv = eng._bufs(ws)['outv'].numpy()
    assert np.abs(outv[:B] - g['d0.r_logit']).max() <= 5e-5
    assert np.abs(outv[B:2 * B] - g['d0.f_logit']).max() <= 5e-5
    assert np.abs(outv[2 * B:] - g['d0.mixed_logit']).max() <= 5e-5
    assert abs(float(stats[3]) - float(g['d0.gp'])) <= 2e-4 * max(1.0, float(g['d0.gp']))
    assert abs(float(stats[0]) - float(g['d0.loss_D'])) <= 5e-4
    Gd = D0.grad_views()
    for n, p in D0.named_parameters():
        ref = max(float(g['d0.gnorm.' + n]), 0.0)                 # -1 in the fixture: the reference left .grad unset
        got = float(Gd[n].double().norm())
        assert abs(got - ref) <= 1e-3 * max(abs(ref), 1e-3), (n, got, ref)
    # the dense forward (reference call signature) agrees with the gather path
    with torch.no_grad():
        dense = D(torch.nn.functional.one_hot(caps, int(g['meta.V'])).float(), obj, mot, gan.attention_mask(caps), alpha)
    assert np.abs(dense.numpy() - g['d0.r_logit']).max() <= 5e-5
    it = dlsg_amd.GanTrainer(G, D, num_D=int(g['meta.num_D']), gan_lambda=float(g['meta.lambda']))
    it.eps_source = lambda k: eps[k]
    res = it.iteration(frames, regions, caps, lens, 1.0)
    assert abs(res['loss_D'] - float(g['loss_D_mean'])) <= 1e-3
    assert abs(res['wasserstein'] - float(g['wasserstein_mean'])) <= 1e-3
    check_post(D.named_parameters(), g, 'dpost.', 5e-5)
    assert abs(res['cap_loss'] - float(g['cap_loss'])) <= 2e-5
    assert abs(res['loss_G'] - float(g['loss_G'])) <= 5e-4
    assert abs(res['total_loss'] - float(g['total_loss'])) <= 5e-5
    Gv = G.grad_views()
    for k, p in G.named_parameters():
        if 'gnorm.' + k in g:
            ref = float(g['gnorm.' + k])
            got = float(Gv[k].double().norm())
            assert abs(got - ref) <= 1e-3 * max(ref, 1e-6) + 1e-6, (k, got, ref)
    check_post(G.named_parameters(), g, 'post.', 2e-5)


def test_gan_lambda_handler_tracks_the_oracle():
    from oracle import gan_ref as GR
    a, b = dlsg_amd.GANLambdaHandler(50, 0.01), GR.GANLambdaHandlerRef(50, 0.01)
    rng = np.random.RandomState(0)
    assert np.allclose(a.decrease_schedule, b.decrease_schedule) and np.allclose(a.increase_schedule, b.increase_schedule)
    for step in range(1, 1500):
        loss = 3.0 + (0.4 if 300 < step < 420 else 0.0) + 0.01 * rng.randn()
        for h in (a, b):
            h.update_gan_lambda(step // 50, step % 50, loss)
        assert a.get_current_lambda() == b.get_current_lambda() and a.state == b.state, step
    assert min(a.cap_list) > 2.9 and len(a.cap_list) == 200


def test_gan_checkpoint_round_trip(tmp_path):
    """run_gun.py:302-310 layout: save -> load into a fresh GanTrainer -> both continue identically."""
    from helpers import load_gan_case

    def mk(args, vocab):
        m = dlsg_amd.CapGnnModel(args, vocab)
        m.set_ops(EmulOps())
        return m
    args, vocab, g, G, D, frames, regions, caps, lens = load_gan_case('gan_msrvtt', mk, dlsg_amd.DiscV2)
    eps = torch.from_numpy(g['eps_gp'])
    it = dlsg_amd.GanTrainer(G, D, num_D=2)
    it.eps_source = lambda k: eps[k]
    it.iteration(frames, regions, caps, lens, 1.0, epoch=0, i=1)
    path = str(tmp_path / '0.pt')
    dlsg_amd.save_checkpoint(path, 0, it)
    ck = torch.load(path, weights_only=False)
    assert sorted(ck.keys()) == ['cap_list', 'epoch', 'model_d_state_dict', 'model_state_dict', 'optimizer_d_state_dict',
                                 'optimizer_state_dict']
    args2, vocab2, _, G2, D2, *_ = load_gan_case('gan_msrvtt', mk, dlsg_amd.DiscV2)
    it2 = dlsg_amd.GanTrainer(G2, D2, num_D=2)
    it2.eps_source = lambda k: eps[k]
    assert dlsg_amd.load_checkpoint(path, it2) == 0
    assert it2.lambda_handler.cap_list == it.lambda_handler.cap_list and it2.trainer.t == it.trainer.t
    a = it.iteration(frames, regions, caps, lens, 1.0, epoch=0, i=2)
    b = it2.iteration(frames, regions, caps, lens, 1.0, epoch=0, i=2)
    assert abs(a['total_loss'] - b['total_loss']) <= 1e-6 and abs(a['loss_D'] - b['loss_D']) <= 1e-6
    for (k, p), (_, q) in zip(G.named_parameters(), G2.named_parameters()):
        assert torch.equal(p.detach(), q.detach()), k
    for (k, p), (_, q) in zip(D.named_parameters(), D2.named_parameters()):
        assert (p.detach() - q.detach()).abs().max().item() <= 1e-7, k


def test_proposal_and_attention_gradients_reach_the_encoder():
    """d(obj_proposals), d(motion_proposals), d(alpha_all) sent into the autograd bridge (models/model.py:36-40 returns them
    graph-attached) against the oracle's autograd."""
    from oracle import torch_ref as R
    net, g, frames, regions, caps, lens, kind = build('small_msvd')
    args, vocab, _, _ = load_case('small_msvd')
    orc = R.CapGnnModelRef(args, vocab).eval()
    orc.load_state_dict(net.state_dict())
    gen = torch.Generator().manual_seed(3)
    w_obj, w_mot = torch.randn(3, 8, 64, generator=gen), torch.randn(3, 8, 64, generator=gen)
    w_al, w_lg = torch.randn(3, 26, 16, generator=gen), torch.randn(3, 26, 50, generator=gen) * 0.1
    grads = []
    for m in (orc, net):
        out, obj, mot, alpha = m(frames, regions, caps, 26, 1.0)
        ((out * w_lg).sum() + (obj * w_obj).sum() + (mot * w_mot).sum() + (alpha * w_al).sum()).backward()
        grads.append({k: p.grad for k, p in m.named_parameters()})
    for k, ref in grads[0].items():
        if ref is None:
            continue
        err = (grads[1][k] - ref).abs().max().item()
        assert err <= 2e-5 + 2e-4 * ref.abs().max().item(), (k, err)


def test_use_glove_initialises_the_word_embedding(tmp_path, monkeypatch):
    """models/layer.py:310-311,352-385: `./data/{dataset}_glove.npy` if present, else built from the GloVe text file (trailing
    comma stripped, unknown words random) and saved; neither file: an error, not a silently ignored flag."""
    from dlsg_amd.config import Vocabulary
    monkeypatch.chdir(tmp_path)
    args = small_args(use_glove=True, word_size=5)
    vocab = Vocabulary()
    for w in ('cat', 'dog,', 'zebra'):
        vocab.add_word(w)
    with pytest.raises(FileNotFoundError):
        dlsg_amd.CapGnnModel(args, vocab)
    os.makedirs('data')
    with open('data/glove.42B.300d.txt', 'w') as f:
        f.write('dog 1 2 3 4 5\ncat 0.5 0.25 0 -1 -2\nhorse 9 9 9 9 9\n')
    net = dlsg_amd.CapGnnModel(args, vocab)
    E_ = net.decoder.word_embed.weight.detach()
    assert E_[vocab('cat')].tolist() == [0.5, 0.25, 0.0, -1.0, -2.0] and E_[vocab('dog,')].tolist() == [1, 2, 3, 4, 5]
    saved = np.load('data/msvd_glove.npy')
    assert saved.shape == (len(vocab), 5) and np.allclose(saved, E_.numpy())
    os.remove('data/glove.42B.300d.txt')
    net2 = dlsg_amd.CapGnnModel(args, vocab)                      # second run: from the .npy, unknown words included
    assert torch.equal(net2.decoder.word_embed.weight.detach(), E_)
    np.save('data/msvd_glove.npy', saved[:-1])
    with pytest.raises(ValueError):
        dlsg_amd.CapGnnModel(args, vocab)
