"""Beam-search inference (Decoder.forward beam branch models/layer.py:449-460, Decoder.beam_step :489-567,
BeamSearch.search models/allennlp_beamsearch.py:51-294 with per_node_beam_size == beam_size, layer.py:346).

The decode step runs on the HIP kernels with all B*k beams as one batch (the reference loops over the k beams,
layer.py:521-551; rows are independent so the values are the same).  The step-invariant tensors (K', V', the
global-feature gates) are expanded once instead of being re-gathered by back-pointer every step; only the four
LSTM states are reordered.  Candidate selection (top-k over the vocabulary and over the k*k continuations) is
bookkeeping on small tensors and uses torch.topk / gather on the device.
"""
import torch

from . import engine as E


def _expand_rows(t, k):
    B = t.shape[0]
    return t.unsqueeze(1).expand(B, k, *t.shape[1:]).reshape(B * k, *t.shape[1:]).contiguous()


@torch.no_grad()
def beam_infer(model, visual_feats, region_feats):
    model.flatten_parameters_()
    ops, dec = model.ops, model.decoder
    ops.extra_flags = model._gemm_flags(False)
    k = dec.beam_size
    seed = model.next_seed()
    sv = {}
    frames = visual_feats.contiguous().float()
    # ---- encoder + step-invariant decoder work on B rows
    if hasattr(model, '_encode'):
        regions = region_feats.contiguous().float()
        obj, mot = model._encode(frames, regions, False, seed, sv)
        mems, sv['dec_gsrc'] = [obj, mot], [obj, mot]
    else:
        B0, T, F = frames.shape
        enc = E.encvis_fwd(ops, model.encoder, 'encoder', frames.view(B0 * T, F), B0, T, sv, False, seed)
        enc = enc.view(B0, T, -1)
        mems, sv['dec_gsrc'] = [enc], [enc]
    s = E.dec_prepare(ops, dec, mems, sv, False, seed)
    B = frames.shape[0]
    V = dec.vocab_size
    L = dec.max_words
    end = dec.vocab('<end>')
    if k > V:
        raise ValueError('Target vocab size (%d) too small relative to per_node_beam_size (%d)' % (V, k))
    # ---- expand the static tensors to B*k rows once
    s['Kp'] = [_expand_rows(x, k) for x in s['Kp']]
    s['Vp'] = [_expand_rows(x, k) for x in s['Vp']]
    s['gq'] = _expand_rows(s['gq'], k)
    R = B * k
    E.dec_alloc(dec, s, frames, R, L)
    ids = s['IDS']
    ids[0].fill_(dec.vocab('<start>'))
    Emb = dec.word_embed.weight
    ops.embed_fwd(Emb, ids[0], s['WE'][0])                       # beam_step applies no word dropout (layer.py:537)
    logp = torch.empty(R, V, dtype=torch.float32, device=frames.device)

    def step(t):
        E.dec_step(ops, dec, s, t, frames, False, seed, R)
        E.dec_logits(ops, dec, s, t, t + 1)
        ops.log_softmax(s['LOGITS'][t], logp)

    step(0)
    start_lp = logp.view(B, k, V)[:, 0]                          # all k rows of a group are identical at step 0
    top_lp, top_cls = start_lp.topk(k)
    if k == 1 and bool((top_cls == end).all()):
        return top_cls, sv['dec_gsrc'][0], sv['dec_gsrc'][-1], []
    last_lp = top_lp
    preds, backs = [top_cls], []
    after_end = torch.full((R, V), float('-inf'), device=frames.device)
    after_end[:, end] = 0.0
    base = (torch.arange(B, device=frames.device) * k).unsqueeze(1)
    state_keys = ['LHP', 'QH', 'QC', 'LC']
    tmp = {key: torch.empty_like(s[key][0]) for key in state_keys}
    for t in range(1, L):
        last = preds[-1].reshape(R)
        if bool((last == end).all()):
            break
        ids[t].copy_(last)
        ops.embed_fwd(Emb, ids[t], s['WE'][t])
        step(t)
        cleaned = torch.where((last == end).unsqueeze(-1), after_end, logp)
        node_lp, node_cls = cleaned.topk(k)
        summed = (node_lp + last_lp.reshape(R, 1)).reshape(B, k * k)
        best_lp, best_idx = summed.topk(k)
        preds.append(node_cls.reshape(B, k * k).gather(1, best_idx))
        last_lp = best_lp
        back = (best_idx / k).type(torch.int64)                   # allennlp_beamsearch.py:242
        backs.append(back)
        rows = (base + back).reshape(R)
        for key in state_keys:                                    # reorder the recurrent state of slot t+1
            ops.gather_rows(s[key][t + 1], rows, tmp[key])
            ops.copy2d(tmp[key], s[key][t + 1])
    if not backs:
        all_preds = preds[0].unsqueeze(2)
    else:
        rec = [preds[-1].unsqueeze(2)]
        cur = backs[-1]
        for t in range(len(preds) - 2, 0, -1):
            rec.append(preds[t].gather(1, cur).unsqueeze(2))
            cur = backs[t - 1].gather(1, cur)
        rec.append(preds[0].gather(1, cur).unsqueeze(2))
        all_preds = torch.cat(list(reversed(rec)), 2)
    best = last_lp.topk(1)[1].squeeze(1)                          # layer.py:456-460
    out = torch.stack([all_preds[i, best[i], :] for i in range(B)])
    if hasattr(model, '_encode'):
        return out, sv['dec_gsrc'][0], sv['dec_gsrc'][1], []
    return out, 0, 0, 0
