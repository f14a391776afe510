"""Generate tests/golden/*.npz by running the REFERENCE model (imported from /root/reference) on CPU.

Run only in the authoring container:  python tests/golden/make_goldens.py
The reference Python never leaves that container; only the arrays written here do.  Nothing in tests/, smoke()
or bench.py reads /root/reference at run time.

The single missing third-party symbol (allennlp.common.checks.ConfigurationError, used by
models/allennlp_beamsearch.py:12) is stubbed before import, as SURVEY.md section 8c describes.
"""
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
REF = '/root/reference'

for name in ('allennlp', 'allennlp.common', 'allennlp.common.checks'):
    sys.modules[name] = types.ModuleType(name)
sys.modules['allennlp.common.checks'].ConfigurationError = type('ConfigurationError', (Exception,), {})
sys.path.insert(0, REF)

import models.model as ref_model          # noqa: E402
import models.sublayer as ref_sub         # noqa: E402
from dlsg_amd.config import make_args, make_vocab, msvd_shaped, msrvtt_shaped   # noqa: E402
from dlsg_amd.synth import synth_state_dict, synth_batch, checksum              # noqa: E402

torch.set_num_threads(8)


def small_args(**kw):
    base = dict(visual_hidden_size=64, region_projected_size=64, query_hidden_size=48, decode_hidden_size=96,
                a_feature_size=40, m_feature_size=72, region_feature_size=32, word_size=20, num_proposals=8,
                num_obj=16, beam_size=5, train_batch_size=3)
    base.update(kw)
    return make_args(**base)


def run_case(tag, args, V, B, seed, store_weights, store_inter, full_logits=True, model_cls='CapGnnModel'):
    vocab = make_vocab(V)
    torch.manual_seed(0)
    net = getattr(ref_model, model_cls)(args, vocab)
    sd = synth_state_dict(net.state_dict(), seed)
    net.load_state_dict(sd, strict=True)
    net.eval()
    frames, regions, caps, lens = synth_batch(args, V, B, seed + 1)
    out = {'meta.V': V, 'meta.B': B, 'meta.seed': seed, 'cap_lens': lens.numpy()}
    if store_weights:
        for k, v in sd.items():
            out['w.' + k] = v.numpy()
        out['frames'] = frames.numpy(); out['regions'] = regions.numpy(); out['captions'] = caps.numpy()
    for k, (s, a) in checksum(sd).items():
        out['ck.' + k] = np.array([s, a])
    out['ck_in'] = np.array([float(frames.double().sum()), float(regions.double().sum()), float(caps.sum())])

    inter = {}
    hooks = []
    if store_inter and model_cls == 'CapGnnModel':
        def grab(name):
            def fn(mod, inp, res):
                inter[name] = (res[0] if isinstance(res, tuple) else res).detach().numpy().copy()
            return fn
        e = net.encoder
        spec = [('obj.v', e.obj_encoder, 'visual_norm'), ('obj.o', e.obj_encoder, 'obj_norm'),
                ('obj.ov', e.obj_encoder, 'obj_visual_norm'), ('obj.psl', e.obj_encoder, 'v2l_layer'),
                ('mot.v', e.motion_encoder, 'visual_norm'), ('mot.o', e.motion_encoder, 'obj_norm'),
                ('mot.ov', e.motion_encoder, 'obj_visual_norm'), ('mot.psl', e.motion_encoder, 'v2l_layer'),
                ('pre.embed', e.motion_pre_encoder, 'linear_embed'), ('pre.lstm', e.motion_pre_encoder, 'lstm'),
                ('pre.lstm_ln', e.motion_pre_encoder, 'layernorm_lstm'),
                ('pre.sa', e.motion_pre_encoder, 'self_attention'), ('pre.out', e.motion_pre_encoder, 'layernorm_sa')]
        for name, parent, attr in spec:
            mod = getattr(parent, attr, None)
            if mod is None:
                continue
            hooks.append(mod.register_forward_hook(grab(name)))

    # --- teacher-forced forward (tf=1.0 -> coin always true), eval mode: deterministic
    with torch.no_grad():
        logits, obj, mot, alpha = net(frames, regions, caps, 26, 1.0)
    for h in hooks:
        h.remove()
    for k, v in inter.items():
        out['i.' + k] = v
    lg = logits.numpy()
    if full_logits:
        out['logits'] = lg
    else:
        top = torch.topk(logits, 8, dim=-1)
        out['logits_top_val'] = top.values.numpy(); out['logits_top_idx'] = top.indices.numpy()
        out['logits_sum'] = logits.double().sum(-1).numpy()
        out['logits_abs_sum'] = logits.double().abs().sum(-1).numpy()
    t2 = torch.topk(logits, 2, dim=-1).values
    out['logit_margin'] = (t2[..., 0] - t2[..., 1]).numpy()
    if model_cls == 'CapGnnModel':
        out['obj_psl'] = obj.numpy(); out['mot_psl'] = mot.numpy(); out['alpha'] = alpha.numpy()

    # --- greedy and beam inference (evaluate.py:67-68 call convention: caption=None)
    with torch.no_grad():
        net.update_beam_size(1)
        out['greedy_ids'] = net(frames, regions, None)[0].numpy()
        net.update_beam_size(5)
        out['beam5_ids'] = net(frames, regions, None)[0].numpy()

    # --- scheduled sampling: coin order pinned by random.seed(12), tf=0.6 (layer.py:432)
    random.seed(12)
    with torch.no_grad():
        ss_logits = net(frames, regions, caps, 26, 0.6)[0]
    random.seed(12)
    out['ss_coins'] = np.array([random.random() < 0.6 for _ in range(26)])
    if full_logits:
        out['ss_logits'] = ss_logits.numpy()
    else:
        out['ss_logits_sum'] = ss_logits.double().sum(-1).numpy()
        top = torch.topk(ss_logits, 8, dim=-1)
        out['ss_logits_top_val'] = top.values.numpy(); out['ss_logits_top_idx'] = top.indices.numpy()

    # --- caller-side train step (run_gun.py:181-198,233-234): loss, grads, one Adam step
    net.zero_grad()
    outs = net(frames, regions, caps, 26, 1.0)[0]
    rows = torch.cat([outs[j][:lens[j]] for j in range(B)], 0).view(-1, V)
    tgt = torch.cat([caps[j][:lens[j]] for j in range(B)], 0).view(-1)
    loss = torch.nn.CrossEntropyLoss()(rows, tgt)
    opt = torch.optim.Adam(net.parameters(), lr=1.6e-4, betas=(0.5, 0.9))
    loss.backward()
    out['loss'] = np.array(loss.item())
    for k, p in net.named_parameters():
        if p.grad is None:
            out['gnone.' + k] = np.array(1)
        else:
            out['gnorm.' + k] = np.array(float(p.grad.double().norm()))
            if store_weights:
                out['g.' + k] = p.grad.numpy().copy()
    opt.step()
    for k, p in net.named_parameters():
        out['post.' + k] = np.array([float(p.detach().double().sum()), float(p.detach().double().abs().sum())])
    path = os.path.join(HERE, tag + '.npz')
    np.savez_compressed(path, **out)
    print(tag, 'loss', loss.item(), 'size %.1f KB' % (os.path.getsize(path) / 1024))


def sa_mask_case():
    """Masked SelfAttention path (models/sublayer.py:70-72)."""
    torch.manual_seed(0)
    m = ref_sub.SelfAttention(32, 32, 16, 0.3, True).eval()
    sd = synth_state_dict(m.state_dict(), 7)
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 26, 32, generator=g)
    valid = torch.tensor([[1.] * 20 + [0.] * 6, [1.] * 9 + [0.] * 17])
    mask = valid.unsqueeze(2) * valid.unsqueeze(1)
    with torch.no_grad():
        y = m(x, mask)
        y0 = m(x)
    out = {'x': x.numpy(), 'mask': mask.numpy(), 'y_masked': y.numpy(), 'y': y0.numpy()}
    for k, v in sd.items():
        out['w.' + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, 'sa_mask.npz'), **out)
    print('sa_mask ok')


if __name__ == '__main__':
    run_case('small_msvd', small_args(), V=50, B=3, seed=11, store_weights=True, store_inter=True)
    run_case('small_msrvtt', small_args(num_obj=6, num_proposals=5, decode_hidden_size=80, dataset='msr-vtt'),
             V=61, B=4, seed=12, store_weights=True, store_inter=True)
    run_case('small_noobj', small_args(num_obj=4), V=50, B=2, seed=13, store_weights=True, store_inter=True)
    run_case('small_baseline1', small_args(), V=50, B=3, seed=14, store_weights=True, store_inter=False,
             model_cls='CapBaseline1')
    run_case('small_baselinemodel', small_args(), V=50, B=3, seed=15, store_weights=True, store_inter=False,
             model_cls='CapBaselineModel')
    sa_mask_case()
    run_case('full_msvd_b2', msvd_shaped(), V=1000, B=2, seed=21, store_weights=False, store_inter=False)
    run_case('full_msrvtt_b2', msrvtt_shaped(), V=10000, B=2, seed=22, store_weights=False, store_inter=False,
             full_logits=False)
