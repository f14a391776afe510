"""Shared helpers for the parity tests: load a golden fixture and rebuild its config / weights / inputs."""
import os

import numpy as np
import torch

from dlsg_amd.config import make_args, make_vocab, msvd_shaped, msrvtt_shaped
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
