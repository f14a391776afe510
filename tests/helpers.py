"""Shared helpers for the parity tests: load a golden fixture and rebuild its config / weights / inputs."""
import os

import numpy as np
import torch

from dlsg_amd.config import make_args, make_vocab, msvd_shaped, msrvtt_shaped, apply_dataset_overrides
from dlsg_amd.synth import synth_state_dict, synth_batch, checksum

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def small_args(**kw):
    base = dict(visual_hidden_size=64, region_projected_size=64, query_hidden_size=48, decode_hidden_size=96,
                a_feature_size=40, m_feature_size=72, region_feature_size=32, word_size=20, num_proposals=8,
                num_obj=16, beam_size=5, train_batch_size=3)
    base.update(kw)
    return make_args(**base)


CASES = {
    'small_msvd': (lambda: small_args(), 'capgnn'),
    'small_msrvtt': (lambda: small_args(num_obj=6, num_proposals=5, decode_hidden_size=80, dataset='msr-vtt'), 'capgnn'),
    'small_noobj': (lambda: small_args(num_obj=4), 'capgnn'),
    'small_baseline1': (lambda: small_args(), 'baseline1'),
    'small_baselinemodel': (lambda: small_args(), 'baselinemodel'),
    'full_msvd_b2': (lambda: msvd_shaped(), 'capgnn'),
    'full_msrvtt_b2': (lambda: msrvtt_shaped(), 'capgnn'),
    # the reference's own default feature widths (utils/opt.py:69-70: A = 1536, M = 1024), tests/golden/make_goldens_r3.py
    'full_default_b2': (lambda: apply_dataset_overrides(make_args(dataset='msvd')), 'capgnn'),
}


def load_case(tag):
    """-> (args, vocab, golden dict, kind)"""
    g = dict(np.load(os.path.join(GOLD, tag + '.npz')))
    mk, kind = CASES[tag]
    args = mk()
    vocab = make_vocab(int(g['meta.V']))
    return args, vocab, g, kind


def weights_and_inputs(model, g, args):
    """Regenerate the seeded weights/inputs of a fixture and verify them against the stored checksums."""
    seed = int(g['meta.seed']); V = int(g['meta.V']); B = int(g['meta.B'])
    sd = synth_state_dict(model.state_dict(), seed)
    for k, (s, a) in checksum(sd).items():
        ref = g['ck.' + k]
        assert abs(s - ref[0]) <= 1e-6 * max(1.0, abs(ref[1])), 'weight RNG drift in ' + k
    frames, regions, caps, lens = synth_batch(args, V, B, seed + 1)
    ck = g['ck_in']
    assert abs(float(frames.double().sum()) - ck[0]) < 1e-3 and float(caps.sum()) == ck[2]
    if 'frames' in g:
        assert np.array_equal(frames.numpy(), g['frames'])
    return sd, frames, regions, caps, torch.as_tensor(g['cap_lens'])


def load_aux(tag, suffix):
    """Round-2 fixtures (tests/golden/make_goldens_r2.py) share config, weights and inputs with the base case `tag`:
    -> golden dict of tests/golden/<tag>_<suffix>.npz"""
    return dict(np.load(os.path.join(GOLD, '%s_%s.npz' % (tag, suffix))))


def check_grads(get_grad, named_parameters, g, rel, abs_=2e-5):
    """every parameter gradient against a fixture holding 'g.<name>' arrays (or 'gnorm.<name>' norms / 'gnone.<name>')"""
    for k, p in named_parameters:
        got = get_grad(k, p)
        if 'gnone.' + k in g:
            assert got is None or float(got.abs().max()) == 0.0, k
        elif 'g.' + k in g:
            ref = g['g.' + k]
            err = np.abs(got.detach().cpu().numpy() - ref).max()
            assert err <= abs_ + rel * np.abs(ref).max(), (k, err, np.abs(ref).max())
        else:
            ref = float(g['gnorm.' + k])
            n = float(got.detach().double().norm())
            assert abs(n - ref) <= 2.5 * rel * max(ref, 1e-6) + 1e-6, (k, n, ref)


def graft_and_step(make_model, device, tmp_path, make_trainer):
    """CapGnnModel.load_encoder (models/model.py:45-53) followed by one optimisation step: returns (model, word-embedding
    before the step, loss).  make_model(args, vocab) -> model; make_trainer(model) -> object with .step(...)."""
    args = small_args()
    vocab = make_vocab(50)
    torch.manual_seed(0)
    donor = make_model(args, vocab).eval()
    sd_donor = synth_state_dict(donor.state_dict(), 31)
    path = os.path.join(str(tmp_path), 'donor.pt')
    torch.save(sd_donor, path)
    net = make_model(args, vocab).eval()
    net.load_state_dict(synth_state_dict(net.state_dict(), 32))
    donor, net = donor.to(device), net.to(device)
    net.load_encoder(donor, path)
    frames, regions, caps, lens = synth_batch(args, 50, 3, 33)
    emb0 = net.decoder.word_embed.weight.detach().cpu().clone()
    assert torch.equal(emb0, sd_donor['decoder.word_embed.weight'])
    tr = make_trainer(net)
    loss = tr.step(frames.to(device), regions.to(device), caps.to(device), lens, 1.0)
    return net, emb0, float(loss)


class OracleTrainer(object):
    """run_gun.py:91,181-198,233-234 around the oracle model: torch.optim.Adam over model.parameters()"""

    def __init__(self, model, lr=1.6e-4):
        from oracle import torch_ref as R
        self.R, self.model = R, model
        self.opt = R.make_optimizer(model, lr)

    def step(self, frames, regions, caps, lens, tf):
        return self.R.train_step(self.model, self.opt, frames, regions, caps, lens, tf)


def gan_args(**kw):
    """tests/golden/make_goldens_r2.py gan_args: DiscV2 hard-codes 1024-wide proposals, the rest stays small"""
    base = dict(visual_hidden_size=1024, region_projected_size=1024, num_topk=3)
    base.update(kw)
    return small_args(**base)


GAN_CASES = {
    'gan_msvd': lambda: gan_args(),
    'gan_msrvtt': lambda: gan_args(num_obj=6, num_proposals=5, num_topk=5, decode_hidden_size=80, dataset='msr-vtt'),
}


def load_gan_case(tag, make_G, make_D):
    """-> (args, vocab, golden, G, D, frames, regions, padded captions, cap_lens); weights regenerated from the seeds"""
    g = dict(np.load(os.path.join(GOLD, tag + '.npz')))
    args = GAN_CASES[tag]()
    V, B, seed = int(g['meta.V']), int(g['meta.B']), int(g['meta.seed'])
    vocab = make_vocab(V)
    torch.manual_seed(0)
    G = make_G(args, vocab)
    G.load_state_dict(synth_state_dict(G.state_dict(), seed), strict=True)
    D = make_D(args, V)
    D.load_state_dict(synth_state_dict(D.state_dict(), seed + 1), strict=True)
    frames, regions, caps, lens = synth_batch(args, V, B, seed + 2)
    for j in range(B):
        caps[j, int(lens[j]):] = 0
    assert np.array_equal(caps.numpy(), g['captions'])
    return args, vocab, g, G.eval(), D.eval(), frames, regions, caps, lens


def check_post(named_parameters, g, prefix, tol):
    """parameters after an Adam step against the fixture's (sum, abs-sum) checksums"""
    for k, p in named_parameters:
        s, a = g[prefix + k]
        got = float(p.detach().double().sum())
        assert abs(got - s) <= tol * max(1.0, a), (k, got, s)
