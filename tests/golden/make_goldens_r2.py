"""Round-2 fixtures, again produced by running the REFERENCE model (imported from /root/reference) on CPU in the
authoring container:  python tests/golden/make_goldens_r2.py

  *_ss.npz       gradients under scheduled sampling (models/layer.py:432-439): eval mode, tf = 0.6, coin order pinned
                 by random.seed(12); loss and every parameter gradient of the caller-side step (run_gun.py:181-198).
  *_drop.npz     TRAIN mode (dropout on).  torch's dropout RNG stream cannot be reproduced on the GPU, so
                 `torch.nn.functional.dropout` -- as seen by the imported reference -- is replaced by a function that takes
                 its keep/scale mask from the build's stateless counter hash (tests/emul_ops.drop_scale ==
                 csrc/common.hpp), keyed by CALL ORDER -> (site, first row).  The reference's own code therefore decides
                 where a dropout sits, on which tensor, with which p; the fixture pins the nine sites
                 (layer.py:28,53,310,320,328; sublayer.py:21-26,58-61,87,183-187) of the build against it.
  gan_*.npz      DiscV2 / WGAN-GP (models/model.py:110-168, run_gun.py:339-398,210-231), see gan_case().
  scoring.json   BLEU / ROUGE_L / CIDEr of caption-eval/pycocoevalcap on synthetic tokenized captions, see scoring_case().

Only arrays leave the container; nothing in tests/, smoke() or bench.py reads /root/reference at run time.
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
sys.path.insert(0, os.path.join(ROOT, 'tests'))
REF = '/root/reference'

for name in ('allennlp', 'allennlp.common', 'allennlp.common.checks'):
    sys.modules[name] = types.ModuleType(name)
sys.modules['allennlp.common.checks'].ConfigurationError = type('ConfigurationError', (Exception,), {})
sys.path.insert(0, REF)

import models.model as ref_model          # noqa: E402
from dlsg_amd.config import make_args, make_vocab, msvd_shaped, msrvtt_shaped   # noqa: E402
from dlsg_amd.synth import synth_state_dict, synth_batch                         # noqa: E402
from dlsg_amd import engine as E                                                 # noqa: E402  (site numbers only)
from emul_ops import drop_scale                                                  # noqa: E402
from make_goldens import small_args                                              # noqa: E402

torch.set_num_threads(8)


def caller_step(net, frames, regions, caps, lens, V, tf):
    """run_gun.py:181-198: forward, ragged CrossEntropy, backward.  Returns (logits, loss)."""
    B = frames.shape[0]
    net.zero_grad()
    outs = net(frames, regions, caps, 26, tf)[0]
    rows = torch.cat([outs[j][:lens[j]] for j in range(B)], 0).view(-1, V)
    tgt = torch.cat([caps[j][:lens[j]] for j in range(B)], 0).view(-1)
    loss = torch.nn.CrossEntropyLoss()(rows, tgt)
    loss.backward()
    return outs.detach(), loss.item()


def store_grads(out, net, full):
    for k, p in net.named_parameters():
        if p.grad is None:
            out['gnone.' + k] = np.array(1)
        else:
            out['gnorm.' + k] = np.array(float(p.grad.double().norm()))
            if full:
                out['g.' + k] = p.grad.numpy().copy()


def ss_case(tag, args, V, B, seed, full, model_cls='CapGnnModel'):
    vocab = make_vocab(V)
    torch.manual_seed(0)
    net = getattr(ref_model, model_cls)(args, vocab)
    net.load_state_dict(synth_state_dict(net.state_dict(), seed), strict=True)
    net.eval()
    frames, regions, caps, lens = synth_batch(args, V, B, seed + 1)
    random.seed(12)
    logits, loss = caller_step(net, frames, regions, caps, lens, V, 0.6)
    random.seed(12)
    out = {'meta.V': V, 'meta.B': B, 'meta.seed': seed, 'cap_lens': lens.numpy(), 'loss': np.array(loss),
           'coins': np.array([random.random() < 0.6 for _ in range(26)])}
    if full:
        out['logits'] = logits.numpy()
    store_grads(out, net, full)
    np.savez_compressed(os.path.join(HERE, tag + '.npz'), **out)
    print(tag, 'loss', loss, 'teacher-forced steps', int(out['coins'].sum()))


def dropout_schedule(B, L, p_model, att_p=0.1):
    """(site, first row, p) of every dropout call of CapGnnModel.forward in train mode, in the reference's call order."""
    sched = [(E.SITE_PSL_OBJ, 0, 0.3),                # obj_encoder.v2l_layer.out_norm          sublayer.py:183-187
             (E.SITE_LSTM, 0, p_model),               # motion_pre_encoder.drop_lstm             layer.py:28,53
             (E.SITE_PE, 0, 0.2),                     # self_attention.pe.dropout                sublayer.py:87-89,104
             (E.SITE_SA, 0, p_model),                 # self_attention.output_layer[1]           sublayer.py:58-61 (layer.py:32)
             (E.SITE_PSL_MOT, 0, 0.3),                # motion_encoder.v2l_layer.out_norm
             (E.SITE_WORD, 0, p_model)]               # word_drop on <start>                     layer.py:310,422
    for t in range(L):
        s = E.STEP_SITE * (t + 1)
        sched += [(s + E.SITE_QUERY, 0, p_model),     # query_lstm_drop                          layer.py:320,574
                  (s + E.SITE_ATT1, 0, att_p),        # context_att.output_layer[3]              sublayer.py:21-26
                  (s + E.SITE_ATT2, 0, att_p),        # context_att_2.output_layer[3]
                  (s + E.SITE_LANG, 0, p_model),      # lang_lstm_drop                           layer.py:328,594
                  (E.SITE_WORD, (t + 1) * B, p_model)]   # word_drop on the next word            layer.py:439
    return sched


class HashDropout(object):
    """Stand-in for torch.nn.functional.dropout: mask of call k = drop_scale(seed, site_k, first_row_k * n + flat index)."""

    def __init__(self, seed, sched):
        self.seed, self.sched, self.k = seed, sched, 0

    def __call__(self, input, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return input
        site, row0, want_p = self.sched[self.k]
        assert abs(p - want_p) < 1e-12, ('dropout call %d: reference p = %g, schedule says %g' % (self.k, p, want_p))
        self.k += 1
        n = input.shape[-1]
        idx = np.arange(input.numel(), dtype=np.uint64) + np.uint64(row0 * n)
        return input * drop_scale(self.seed, site, idx, p).view(input.shape)


def model_seed(counter):
    """dlsg_amd.model._HipModel.next_seed() for seed_counter == counter"""
    return (0x5DEECE66D * counter + 0xB) & 0xFFFFFFFFFFFF


def drop_case(tag, args, V, B, seed, tf, counter=7):
    import torch.nn.functional as F
    vocab = make_vocab(V)
    torch.manual_seed(0)
    net = ref_model.CapGnnModel(args, vocab)
    net.load_state_dict(synth_state_dict(net.state_dict(), seed), strict=True)
    net.train()
    frames, regions, caps, lens = synth_batch(args, V, B, seed + 1)
    hd = HashDropout(model_seed(counter), dropout_schedule(B, 26, args.dropout))
    orig = F.dropout
    F.dropout = hd
    try:
        random.seed(4)
        logits, loss = caller_step(net, frames, regions, caps, lens, V, tf)
    finally:
        F.dropout = orig
    assert hd.k == len(hd.sched), (hd.k, len(hd.sched))
    random.seed(4)
    out = {'meta.V': V, 'meta.B': B, 'meta.seed': seed, 'meta.counter': counter, 'meta.tf': tf, 'cap_lens': lens.numpy(),
           'loss': np.array(loss), 'logits': logits.numpy(), 'coins': np.array([random.random() < tf for _ in range(26)])}
    store_grads(out, net, True)
    np.savez_compressed(os.path.join(HERE, tag + '.npz'), **out)
    print(tag, 'loss', loss, 'dropout calls', hd.k)


def gan_args(**kw):
    """DiscV2 hard-codes 1024-wide proposals and 512-wide internals (models/layer.py:665-667): H = 1024, the rest small."""
    base = dict(visual_hidden_size=1024, region_projected_size=1024, num_topk=3)
    base.update(kw)
    return small_args(**base)


def gan_case(tag, args, V, B, seed, num_D=5, gan_lambda=0.01):
    """One RunGAN iteration (run_gun.py:153-234, 339-381) around the REFERENCE CapGnnModel and DiscV2, both in eval mode
    (dropout off), tf = 1.0; `torch.rand` of the gradient penalty replaced by recorded numbers."""
    from dlsg_amd.synth import checksum
    vocab = make_vocab(V)
    torch.manual_seed(0)
    G = ref_model.CapGnnModel(args, vocab)
    G.load_state_dict(synth_state_dict(G.state_dict(), seed), strict=True)
    D = ref_model.DiscV2(args, V)
    D.load_state_dict(synth_state_dict(D.state_dict(), seed + 1), strict=True)
    G.eval(); D.eval()
    frames, regions, caps, lens = synth_batch(args, V, B, seed + 2)
    for j in range(B):
        caps[j, int(lens[j]):] = 0                      # <pad> beyond the caption, as the loader delivers it
    out = {'meta.V': V, 'meta.B': B, 'meta.seed': seed, 'meta.num_D': num_D, 'meta.lambda': gan_lambda, 'cap_lens': lens.numpy(),
           'captions': caps.numpy()}
    max_len, bs = 26, B
    opt_G = torch.optim.Adam(G.parameters(), lr=1.6e-4, betas=(0.5, 0.9))
    opt_D = torch.optim.Adam(D.parameters(), lr=1.6e-4, betas=(0.5, 0.9))
    targets = caps
    # ---- Train D (run_gun.py:163-178)
    seq_mask = (caps > 0).to(torch.float32)
    att_mask = torch.matmul(seq_mask.view(bs, max_len, 1), seq_mask.view(bs, 1, max_len))
    f_caption, object_psl, motion_psl, alpha_all = G(frames, regions, targets, max_len, 1.0)
    f_caption = f_caption.detach()
    r_caption = torch.zeros(bs, max_len, V).scatter_(2, targets.unsqueeze(2), 1)          # to_onehot, run_gun.py:447-451
    object_psl, motion_psl, alpha_all = object_psl.detach(), motion_psl.detach(), alpha_all.detach()
    gen = torch.Generator().manual_seed(777)
    eps_all = torch.rand(num_D, bs, 1, 1, generator=gen)
    out['eps_gp'] = eps_all.numpy()
    mean_loss, mean_w = 0.0, 0.0
    for k in range(num_D):                               # run_gun.py:343-381
        opt_D.zero_grad()
        r_logit = D(r_caption, object_psl, motion_psl, att_mask, alpha_all)
        f_logit = D(f_caption, object_psl, motion_psl, att_mask, alpha_all)
        epsilon_gp = eps_all[k].clone().requires_grad_(True)
        mixed = r_caption.detach() * epsilon_gp + f_caption.detach() * (1 - epsilon_gp)
        mixed_logit = D(mixed, object_psl, motion_psl, att_mask, alpha_all)
        grad_gp = torch.autograd.grad(inputs=mixed, outputs=mixed_logit, grad_outputs=torch.ones_like(mixed_logit),
                                      create_graph=True, retain_graph=True)[0]
        gnorm = grad_gp.contiguous().view(len(grad_gp), -1).norm(2, dim=1)
        gp = ((gnorm - 1) * (gnorm - 1)).mean()
        r_loss, f_loss = r_logit.mean(), f_logit.mean()
        loss_D = f_loss - r_loss + 10 * gp
        mean_loss += loss_D.item() / num_D
        mean_w += (r_loss.item() - f_loss.item()) / num_D
        loss_D.backward(retain_graph=True)
        if k == 0:
            out.update({'d0.r_logit': r_logit.detach().numpy(), 'd0.f_logit': f_logit.detach().numpy(),
                        'd0.mixed_logit': mixed_logit.detach().numpy(), 'd0.grad_norm': gnorm.detach().numpy(),
                        'd0.gp': np.array(gp.item()), 'd0.loss_D': np.array(loss_D.item())})
            for n, p in D.named_parameters():
                out['d0.gnorm.' + n] = np.array(float(p.grad.double().norm())) if p.grad is not None else np.array(-1.0)
        opt_D.step()
    out['loss_D_mean'] = np.array(mean_loss); out['wasserstein_mean'] = np.array(mean_w)
    for n, (s_, a_) in checksum(dict(D.named_parameters())).items():
        out['dpost.' + n] = np.array([s_, a_])
    # ---- Train the captioning model (run_gun.py:180-234)
    opt_G.zero_grad()
    outputs, object_psl, motion_psl, alpha_all = G(frames, regions, targets, max_len, 1.0)
    tokens = outputs
    rows = torch.cat([outputs[j][:lens[j]] for j in range(bs)], 0).view(-1, V)
    tgt = torch.cat([targets[j][:lens[j]] for j in range(bs)], 0).view(-1)
    cap_loss = torch.nn.CrossEntropyLoss()(rows, tgt)
    f_logit = D(tokens, object_psl.detach(), motion_psl.detach(), att_mask=att_mask, alpha_all=alpha_all.detach())
    loss_G = -f_logit.mean()
    total = cap_loss + loss_G * gan_lambda
    total.backward()
    out.update({'cap_loss': np.array(cap_loss.item()), 'loss_G': np.array(loss_G.item()), 'total_loss': np.array(total.item()),
                'g.f_logit': f_logit.detach().numpy()})
    for n, p in G.named_parameters():
        if p.grad is None:
            out['gnone.' + n] = np.array(1)
        else:
            out['gnorm.' + n] = np.array(float(p.grad.double().norm()))
    opt_G.step()
    for n, (s_, a_) in checksum(dict(G.named_parameters())).items():
        out['post.' + n] = np.array([s_, a_])
    np.savez_compressed(os.path.join(HERE, tag + '.npz'), **out)
    print(tag, 'loss_D', mean_loss, 'wasserstein', mean_w, 'cap', cap_loss.item(), 'loss_G', loss_G.item(), 'gp0', float(out['d0.gp']))


def scoring_case():
    """BLEU / ROUGE_L / CIDEr of the reference's own scorer classes (caption-eval/pycocoevalcap, pure Python) on synthetic
    tokenized captions; inputs and outputs go to scoring.json."""
    import json
    sys.path.insert(0, os.path.join(REF, 'caption-eval'))
    from pycocoevalcap.bleu.bleu import Bleu
    from pycocoevalcap.rouge.rouge import Rouge
    from pycocoevalcap.cider.cider import Cider
    rng = np.random.RandomState(5)
    words = ['a', 'man', 'woman', 'dog', 'cat', 'is', 'are', 'playing', 'running', 'cooking', 'the', 'guitar', 'ball', 'with',
             'in', 'kitchen', 'park', 'on', 'street', 'two', 'people', 'riding', 'horse', 'bike', 'slicing', 'onion', 'water']
    cases = []
    for nvid, nref in ((12, 5), (40, 17), (3, 1)):
        gts, res = {}, {}
        for v in range(nvid):
            base = [words[i] for i in rng.randint(0, len(words), size=rng.randint(3, 12))]
            refs = []
            for _ in range(nref):
                r = list(base)
                for _ in range(rng.randint(0, 4)):
                    r[rng.randint(0, len(r))] = words[rng.randint(0, len(words))]
                if rng.rand() < 0.3:
                    r = r[:max(2, len(r) - 2)]
                refs.append(' '.join(r))
            h = list(base)
            for _ in range(rng.randint(0, 5)):
                h[rng.randint(0, len(h))] = words[rng.randint(0, len(words))]
            if v == 1:
                h = ['zebra']                                   # no overlap at all
            gts['vid%d' % v] = refs
            res['vid%d' % v] = [' '.join(h)]
        b, bs = Bleu(4).compute_score(gts, res)
        r, rs = Rouge().compute_score(gts, res)
        c, cs = Cider().compute_score(gts, res)
        cases.append({'gts': gts, 'res': res, 'bleu': [float(x) for x in b], 'bleu_per': [[float(y) for y in x] for x in bs],
                      'rouge': float(r), 'rouge_per': [float(x) for x in rs], 'cider': float(c), 'cider_per': [float(x) for x in cs]})
        print('scoring case', nvid, nref, 'BLEU4 %.4f ROUGE %.4f CIDEr %.4f' % (b[3], r, c))
    with open(os.path.join(HERE, 'scoring.json'), 'w') as f:
        json.dump(cases, f)


if __name__ == '__main__':
    which = sys.argv[1:] or ['ss', 'drop', 'gan', 'scoring']
    if 'ss' in which:
        ss_case('small_msvd_ss', small_args(), V=50, B=3, seed=11, full=True)
        ss_case('small_msrvtt_ss', small_args(num_obj=6, num_proposals=5, decode_hidden_size=80, dataset='msr-vtt'),
                V=61, B=4, seed=12, full=True)
        ss_case('small_baseline1_ss', small_args(), V=50, B=3, seed=14, full=True, model_cls='CapBaseline1')
        ss_case('full_msvd_b2_ss', msvd_shaped(), V=1000, B=2, seed=21, full=False)
    if 'drop' in which:
        drop_case('small_msvd_drop', small_args(), V=50, B=3, seed=11, tf=0.8)
        drop_case('small_msrvtt_drop', small_args(num_obj=6, num_proposals=5, decode_hidden_size=80, dataset='msr-vtt'),
                  V=61, B=4, seed=12, tf=1.0)
    if 'scoring' in which:
        scoring_case()
    if 'gan' in which:
        gan_case('gan_msvd', gan_args(), V=50, B=3, seed=51)
        # num_proposals <= num_topk: PSLScore2 keeps every proposal (layer.py:686-688), MSR-VTT setting of run_gun.py:36-40
        gan_case('gan_msrvtt', gan_args(num_obj=6, num_proposals=5, num_topk=5, decode_hidden_size=80, dataset='msr-vtt'),
                 V=61, B=4, seed=53)
