"""Deterministic synthetic weights and inputs (there are no datasets or checkpoints on the GPU box).

`synth_state_dict` fills any module's state_dict from per-tensor seeded CPU generators, so the authoring
container (which loads the result into the imported reference model to make tests/golden/*.npz) and the GPU
box (which loads it into the HIP model) see bit-identical weights.  Input recipe follows SURVEY.md section 8d.
"""
import math

import torch


def synth_state_dict(template, seed=0):
    """template: a state_dict (name -> tensor).  Returns a new dict with the same keys/shapes/dtypes.

    >=2-d tensors : U(-1,1)*sqrt(3/fan_in)  (unit-variance preserving);  embeddings U(-1,1)
    1-d '*weight' : 1 + 0.2*U(-1,1)  (LayerNorm gain)
    1-d others    : 0.2*U(-1,1)      (biases)
    buffers named '*.pe' are kept (deterministic sinusoid table).
    """
    out = {}
    for i, (k, t) in enumerate(template.items()):
        if k.endswith('.pe'):
            out[k] = t.clone()
            continue
        g = torch.Generator().manual_seed(seed * 100003 + i)
        u = torch.rand(t.shape, generator=g, dtype=torch.float32) * 2 - 1
        if t.dim() >= 2:
            if 'word_embed' in k:
                w = u
            elif t.dim() == 3:          # Conv1d weight (out, in, k) of the DiscV2 critic: fan-in = in * k
                w = u * math.sqrt(3.0 / (t.shape[1] * t.shape[2]))
            else:
                w = u * math.sqrt(3.0 / t.shape[-1])
        elif k.endswith('weight'):
            w = 1 + 0.2 * u
        else:
            w = 0.2 * u
        out[k] = w.to(t.dtype)
    return out


def checksum(state):
    """name -> (sum, abs-sum) in float64; stored in fixtures to detect RNG drift across torch builds."""
    return {k: (float(v.double().sum()), float(v.double().abs().sum())) for k, v in state.items()}


def synth_batch(args, vocab_size, batch, seed=1, device='cpu'):
    """frames N(0,1) (B,T,A+M); regions N(0,1) (B,T,O,R); captions U{4..V-1} (B,26) int64; cap_lens U{5..26}
    sorted descending (the reference collate sorts by length, utils/data.py:90)."""
    g = torch.Generator().manual_seed(seed)
    T = args.max_frames
    frames = torch.randn(batch, T, args.a_feature_size + args.m_feature_size, generator=g)
    regions = torch.randn(batch, T, args.num_obj, args.region_feature_size, generator=g)
    captions = torch.randint(4, vocab_size, (batch, args.max_words), generator=g)
    lens = torch.randint(5, args.max_words + 1, (batch,), generator=g).sort(descending=True)[0]
    return frames.to(device), regions.to(device), captions.to(device), lens
