"""SURVEY.md section 8 row f3 against fixtures the REFERENCE's own code wrote (tests/golden/make_goldens_r4.py):
`utils.utils.GANLambdaHandler` traced over 2 400 steps with a checkpoint-style resume, and a checkpoint dict written by
`torch.save` from the reference's models and optimizers (run_gun.py:302-310).  Oracle and product are both held to them."""
import lzma
import os

import numpy as np
import pytest
import torch

import dlsg_amd
from dlsg_amd.config import make_vocab
from dlsg_amd.synth import checksum
from emul_ops import EmulOps
from helpers import GOLD, small_args


def _handlers():
    from oracle import gan_ref as GR
    return [('oracle', GR.GANLambdaHandlerRef), ('product', dlsg_amd.GANLambdaHandler)]


@pytest.mark.parametrize('which', ['oracle', 'product'])
def test_lambda_handler_follows_the_reference_trace(which):
    """utils/utils.py:196-265 over stable -> rise -> dip (cut by a resume that keeps only cap_list, run_gun.py:101-109) ->
    stable -> second rise -> full 500-step dip -> stable: lambda, state, schedule position and cap_list at every step."""
    cls = dict(_handlers())[which]
    g = np.load(os.path.join(GOLD, 'lambda_trace.npz'))
    total, lam0, resume_at = int(g['meta.total_step']), float(g['meta.gan_lambda']), int(g['meta.resume_at'])
    h = cls(total, lam0)
    assert np.allclose(h.decrease_schedule, g['decrease_schedule'], rtol=0, atol=1e-15)
    assert np.allclose(h.increase_schedule, g['increase_schedule'], rtol=0, atol=1e-15)
    seen = set()
    for k in range(len(g['loss'])):
        if k == resume_at:
            assert np.array_equal(np.array(h.cap_list), g['resume.cap_list'])
            h = cls(total, lam0, cap_list=g['resume.cap_list'])
        h.update_gan_lambda(int(g['epoch'][k]), int(g['i'][k]), float(g['loss'][k]))
        lam = h.get_current_lambda()
        assert abs(lam - g['lam'][k]) <= 1e-15, (k, lam, g['lam'][k])
        assert h.state == g['state'][k] and h.current_schedule_step == g['sched_step'][k], k
        assert h.current_step == g['cur_step'][k] and len(h.cap_list) == g['n_cap'][k], k
        assert abs(float(np.sum(h.cap_list)) - g['cap_sum'][k]) <= 1e-9, k
        seen.add((int(g['state'][k - 1]), int(g['state'][k])) if k else (0, 0))
    assert {(0, 1), (1, 0), (1, 1), (0, 0)} <= seen


def two_level(shape, seed, mag):
    """the gradient recipe of make_goldens_r4.py (data: random signs times a magnitude)"""
    g = torch.Generator().manual_seed(seed)
    return (torch.randint(0, 2, tuple(shape), generator=g).to(torch.float32) * 2 - 1) * mag


def _unpack(tmp_path):
    path = os.path.join(str(tmp_path), 'ref_checkpoint.pt')
    with lzma.open(os.path.join(GOLD, 'ref_checkpoint.pt.xz'), 'rb') as f, open(path, 'wb') as out:
        out.write(f.read())
    return path


def _same_adam_state(got, want, names):
    assert set(int(k) for k in got['state']) == set(int(k) for k in want['state'])
    for k, st in want['state'].items():
        mine = got['state'][k]
        assert float(mine['step']) == float(st['step']), names[int(k)]
        for f in ('exp_avg', 'exp_avg_sq'):
            assert torch.equal(mine[f].cpu(), st[f]), (names[int(k)], f)
    a, b = got['param_groups'][0], want['param_groups'][0]
    assert a['lr'] == b['lr'] and tuple(a['betas']) == tuple(b['betas']) and a['eps'] == b['eps'] and a['params'] == b['params']


def _third_step_matches(named, nxt, prefix):
    for n, p in named:
        s, a = nxt[prefix + n]
        got = float(p.detach().double().sum())
        assert abs(got - s) <= 2e-6 * max(1.0, a), (n, got, s)


def test_oracle_resumes_from_the_reference_checkpoint(tmp_path):
    from oracle import torch_ref as R, gan_ref as GR
    path = _unpack(tmp_path)
    ck = torch.load(path, map_location='cpu', weights_only=False)
    nxt = np.load(os.path.join(GOLD, 'ref_checkpoint_next.npz'))
    V = int(nxt['meta.V'])
    args = small_args(num_topk=3)
    G, D = R.CapGnnModelRef(args, make_vocab(V)), GR.DiscV2Ref(args, V)
    G.load_state_dict(ck['model_state_dict'], strict=True)
    D.load_state_dict(ck['model_d_state_dict'], strict=True)
    opt_G = torch.optim.Adam(G.parameters(), lr=1.0, betas=(0.5, 0.9))
    opt_D = torch.optim.Adam(D.parameters(), lr=1.0, betas=(0.5, 0.9))
    opt_G.load_state_dict(ck['optimizer_state_dict'])
    opt_D.load_state_dict(ck['optimizer_d_state_dict'])
    none_G = set(nxt['none_G'].tolist())
    for net, off in ((G, 0), (D, 4000)):
        for j, (n, p) in enumerate(net.named_parameters()):
            p.grad = None if (net is G and n in none_G) else two_level(p.shape, 9000 * 3 + j + off, 0.75)
    opt_G.step(); opt_D.step()
    _third_step_matches(G.named_parameters(), nxt, 'G.')
    _third_step_matches(D.named_parameters(), nxt, 'D.')
    h = GR.GANLambdaHandlerRef(50, 0.01, cap_list=ck['cap_list'])
    assert len(h.cap_list) == 200 and h.cap_list == list(ck['cap_list'])


def _product_resume(tmp_path, device, ops):
    path = _unpack(tmp_path)
    ck = torch.load(path, map_location='cpu', weights_only=False)
    nxt = np.load(os.path.join(GOLD, 'ref_checkpoint_next.npz'))
    V = int(nxt['meta.V'])
    args = small_args(num_topk=3)
    G = dlsg_amd.CapGnnModel(args, make_vocab(V))
    # the reference's DiscV2 ignores visual_hidden_size (its 1024 / 512 widths are literals); the product's constructor checks it
    D = dlsg_amd.DiscV2(small_args(num_topk=3, visual_hidden_size=1024), V)
    if ops is not None:
        G.set_ops(ops); D.set_ops(ops)
    G, D = G.to(device), D.to(device)
    gan = dlsg_amd.GanTrainer(G, D, lr=1.0)                 # lr must come from the checkpoint's param_groups
    assert dlsg_amd.load_checkpoint(path, gan) == 4
    for k, v in ck['model_state_dict'].items():
        assert torch.equal(G.state_dict()[k].cpu(), v), k
    for k, v in ck['model_d_state_dict'].items():
        assert torch.equal(D.state_dict()[k].cpu(), v), k
    assert gan.trainer.t == 2 and gan.trainer.lr == ck['optimizer_state_dict']['param_groups'][0]['lr']
    _same_adam_state(gan.trainer.optimizer_state_dict(), ck['optimizer_state_dict'], [n for n, _ in G.named_parameters()])
    _same_adam_state(gan.optimizer_d_state_dict(), ck['optimizer_d_state_dict'], [n for n, _ in D.named_parameters()])
    assert gan.lambda_handler.cap_list == list(ck['cap_list'])
    # the reference's continuation: a third Adam step of both optimizers on the recorded gradients
    none_G = set(nxt['none_G'].tolist())
    Gv = G.grad_views()
    for j, (n, p) in enumerate(G.named_parameters()):
        if n in none_G:
            Gv[n].zero_()
        else:
            Gv[n].copy_(two_level(p.shape, 9000 * 3 + j, 0.75))
    gan.trainer.t += 1
    gan.trainer._adam(gan.trainer.t)
    gan.critic_adam_step({n: two_level(p.shape, 9000 * 3 + j + 4000, 0.75).to(device) for j, (n, p) in enumerate(D.named_parameters())})
    if device != 'cpu':
        torch.cuda.synchronize()
    _third_step_matches(G.named_parameters(), nxt, 'G.')
    _third_step_matches(D.named_parameters(), nxt, 'D.')
    return gan


def test_product_resumes_from_the_reference_checkpoint(tmp_path):
    _product_resume(tmp_path, 'cpu', EmulOps())


@pytest.mark.gpu
def test_product_resumes_from_the_reference_checkpoint_on_the_gpu(tmp_path):
    _product_resume(tmp_path, 'cuda', None)
