"""Beam-search inference (Decoder.forward beam branch models/layer.py:449-460, Decoder.beam_step :489-567,
BeamSearch.search models/allennlp_beamsearch.py:51-294 with per_node_beam_size == beam_size, layer.py:346).

The decode step runs on the HIP kernels with all B*k beams as one batch (the reference loops over the k beams,
layer.py:521-551; rows are independent so the values are the same).  The step-invariant tensors are never re-gathered by
back-pointer: K', V' stay one block per clip that its k beams read (dlsg_dec_mid_args.kv_div), the global-feature gates
are expanded once; only the four LSTM states are reordered (one gather launch, ping-pong slots).  Candidate selection -- log-softmax, top-k over the
vocabulary per beam, top-k over the k*k continuations, back-pointers -- is one HIP launch per step (`beam_select`);
the host does not synchronise inside the loop except for the early-exit test every 4th step.
"""
import torch

from . import engine as E


def _expand_rows(t, k):
    B = t.shape[0]
    return t.unsqueeze(1).expand(B, k, *t.shape[1:]).reshape(B * k, *t.shape[1:]).contiguous()


@torch.no_grad()
def beam_infer(model, visual_feats, region_feats):
    """Beam search as the reference runs it (early exit tested every 4th step on the host)."""
    return beam_finish(model, *beam_device(model, visual_feats, region_feats, early_exit=True))


@torch.no_grad()
def beam_device(model, visual_feats, region_feats, early_exit=True):
    """Everything of the search that runs on the device.  early_exit=False: no host synchronisation at all (all L steps
    are launched; beam_finish cuts the result to the step the reference would have stopped at) -- this form is what
    BeamGraph captures into a hipGraph.  Returns the tensors beam_finish needs."""
    model.flatten_parameters_()
    ops, dec = model.ops, model.decoder
    model._gemm_policy(False)
    k = dec.beam_size
    seed = model.next_seed()
    sv = {}
    frames = visual_feats.contiguous().float()
    # ---- encoder + step-invariant decoder work on B rows
    if hasattr(model, '_encode'):
        regions = region_feats.contiguous().float()
        obj, mot = model._encode(frames, regions, False, seed, sv)
        mems, sv['dec_gsrc'] = [obj, mot], [obj, mot]
    elif hasattr(model, '_motion_nodes'):
        mot = model._motion_nodes(frames, region_feats.contiguous().float(), False, seed, sv)
        mems, sv['dec_gsrc'] = [mot], [mot]
    else:
        B0, T, F = frames.shape
        enc = E.encvis_fwd(ops, model.encoder, 'encoder', frames.view(B0 * T, F), B0, T, sv, False, seed)
        enc = enc.view(B0, T, -1)
        mems, sv['dec_gsrc'] = [enc], [enc]
    extras = (sv['dec_gsrc'][0], sv['dec_gsrc'][-1]) if hasattr(model, '_encode') else None
    return beam_search_from(model, mems, sv, seed, early_exit, extras)


@torch.no_grad()
def beam_search_from(model, mems, sv, seed, early_exit=True, extras=None):
    """The search itself from given attended memories (`mems`: the proposals of the encoder streams, or whatever a caller of
    `Decoder.forward` hands in -- models/layer.py:449-455 builds its start state from `cnn_feats` / `cnn_feats_2` / `global_feat`
    the same way); `sv['dec_gsrc']` are the tensors whose row means form the global feature, `sv['step_feats']` replaces them
    (layer.py:404-405)."""
    ops, dec = model.ops, model.decoder
    k = dec.beam_size
    frames = mems[0]                                   # (reference tensor for device / dtype of the scratch arrays)
    s = E.dec_prepare(ops, dec, mems, sv, False, seed)
    B = frames.shape[0]
    V = dec.vocab_size
    L = dec.max_words
    end = dec.vocab('<end>')
    if k > V:
        raise ValueError('Target vocab size (%d) too small relative to per_node_beam_size (%d)' % (V, k))
    # ---- K', V' stay one block per CLIP: the k beams of a clip are consecutive rows and read their clip's block (kv_div) -- no
    # expanded (B*k)-row copies (82 MB at 128 x 5 rows).  Measured: the word step is not faster for it (20.0 ms per batch either
    # way: the fused step kernel is bound by its per-row chain, the expanded copies were Infinity-Cache resident); it saves the
    # memory and the one-time copies.  Only the small global-feature gate term is expanded to B*k rows
    s['kv_div'] = k
    s['gq'] = _expand_rows(s['gq'], k)
    R = B * k
    E.dec_alloc(dec, s, frames, R, L)
    dev = frames.device
    # recurrent state: two physical slots per step.  Step t reads slot 2t and writes slot 2t+1; the reorder by
    # back-pointer gathers slot 2t+1 into slot 2t+2, so nothing is gathered in place and nothing is copied back.
    state_keys = ['LHP', 'QH', 'QC', 'LC']
    big = {key: torch.zeros(2 * L + 2, R, s[key].shape[2], dtype=torch.float32, device=dev) for key in state_keys}
    Emb = dec.word_embed.weight
    preds = torch.empty(L, R, dtype=torch.int64, device=dev)       # chosen classes per step, (B,k) flattened
    backs = torch.zeros(L, R, dtype=torch.int64, device=dev)
    rows = torch.empty(R, dtype=torch.int64, device=dev)
    lps = torch.zeros(2, R, dtype=torch.float32, device=dev)       # running log-probs, ping-pong
    ended = torch.zeros(L, dtype=torch.int32, device=dev)          # number of <end> among the classes chosen at step t
    start = torch.full((R,), dec.vocab('<start>'), dtype=torch.int64, device=dev)

    def step(t, words):
        for key in state_keys:
            s[key] = big[key][t:]                                  # index t -> slot 2t, index t+1 -> slot 2t+1
        ops.embed_fwd(Emb, words, s['WE'][t])                      # beam_step applies no word dropout (layer.py:537)
        E.dec_step(ops, dec, s, t, frames, False, seed, R)
        E.dec_logits(ops, dec, s, t, t + 1)

    # the reference tests `all beams ended` on the host before every step (allennlp_beamsearch.py:168); here the count of
    # <end> tokens is kept on the device and read every CHECK steps, and the result is cut to the step the reference
    # would have stopped at (steps after that only append <end> at log-prob 0 and change nothing before them)
    CHECK = 4
    step(0, start)
    ops.beam_select(s['LOGITS'][0], start, lps[0], preds[0], lps[1], backs[0], rows, k, end, first=True, ended_count=ended[0:1])
    done = 1
    for t in range(1, L):
        if early_exit and t % CHECK == 0:
            cnt = ended[:t].tolist()
            if any(c == R for c in cnt):
                break
        # state of step t: slot 2t <- slot 2t-1, rows reordered by the parents chosen at step t-1
        ops.gather_rows_multi([big[key][2 * t - 1] for key in state_keys], rows, [big[key][2 * t] for key in state_keys])
        step(t, preds[t - 1])
        cur, nxt = lps[t % 2], lps[(t + 1) % 2]
        ops.beam_select(s['LOGITS'][t], preds[t - 1], cur, preds[t], nxt, backs[t], rows, k, end, ended_count=ended[t:t + 1])
        done = t + 1
    return preds, backs, lps, ended, done, B, k, R, L, extras


def beam_finish(model, preds, backs, lps, ended, done, B, k, R, L, extras):
    """Host side of the search: where the reference would have stopped (allennlp_beamsearch.py:168), the back-trace
    (:272-292) and the choice of the best beam (layer.py:456-460)."""
    cnt = ended[:done].tolist()                                   # host synchronisation
    chk = getattr(model.ops, 'check_persistent', None)
    if chk is not None:
        chk()                                                     # the encoder's persistent BiLSTM: time-out word is final here
    n_steps = done
    hit = [i for i, c in enumerate(cnt[:L - 1]) if c == R]
    if hit:
        n_steps = min(done, hit[0] + 1)
    if k == 1 and cnt[0] == R:
        n_steps = 1
    # ---- back-trace over the steps the reference would have run (allennlp_beamsearch.py:272-292)
    P = preds[:n_steps].view(n_steps, B, k)
    Bk = backs[:n_steps].view(n_steps, B, k)
    # once every beam has ended, further steps keep the log-probs and their order, so the last buffer is the reference's
    last_lp = lps[done % 2].view(B, k)
    if n_steps == 1:
        all_preds = P[0].unsqueeze(2)
    else:
        rec = [P[n_steps - 1].unsqueeze(2)]
        cur = Bk[n_steps - 1]
        for t in range(n_steps - 2, 0, -1):
            rec.append(P[t].gather(1, cur).unsqueeze(2))
            cur = Bk[t].gather(1, cur)
        rec.append(P[0].gather(1, cur).unsqueeze(2))
        all_preds = torch.cat(list(reversed(rec)), 2)
    best = last_lp.topk(1)[1].squeeze(1)                          # layer.py:456-460
    # (one gather: indexing clip by clip reads best[i] back to the host B times -- 128 synchronisations per batch)
    out = all_preds.gather(1, best.view(B, 1, 1).expand(B, 1, all_preds.shape[2])).squeeze(1)
    if extras is not None:
        return out, extras[0], extras[1], []
    return out, 0, 0, 0
