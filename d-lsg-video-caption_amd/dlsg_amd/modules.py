"""Parameter containers of the hot path.

These nn.Modules only HOLD parameters/buffers, under the reference's attribute names so that
`state_dict()` keys/shapes are identical to the reference's and a reference checkpoint loads
with strict=True.  They are never called: all arithmetic is done by the HIP kernels driven from engine.py.
Construction order follows the reference constructors so default initialisation draws the same RNG stream
(models/layer.py:8-37,140-170,277-346; models/sublayer.py:11-26,47-61,177-187; models/model.py:26-30,57-67).
"""
import math

import torch
import torch.nn as nn


def sinusoid_pe(d_model, max_len=72):
    """PositionalEncoding_old table (models/sublayer.py:91-98)."""
    pe = torch.zeros(max_len, d_model)
    pos = torch.arange(0., max_len).unsqueeze(1)
    div = torch.exp(torch.arange(0., d_model, 2) * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.unsqueeze(0)


class _PE(nn.Module):
    def __init__(self, d_model):
        super().__init__()
        self.register_buffer('pe', sinusoid_pe(d_model))


class AttentionShare(nn.Module):
    def __init__(self, value_size, key_size, out_size, dropout=0.1):
        super().__init__()
        self.attention_size = out_size
        self.dropout = dropout
        self.K = nn.Linear(value_size, out_size, bias=False)
        self.Q = nn.Linear(key_size, out_size, bias=False)
        self.V = nn.Linear(value_size, out_size, bias=False)
        self.output_layer = nn.Sequential(nn.Linear(out_size, out_size, bias=False), nn.Tanh(),
                                          nn.LayerNorm(out_size), nn.Dropout(dropout))


class SelfAttention(nn.Module):
    def __init__(self, input_size, attention_size, output_size, dropout=0.2, get_pe=False):
        super().__init__()
        self.attention_size = attention_size
        self.dropout = dropout
        self.get_pe = get_pe
        self.pe = _PE(attention_size)
        self.K = nn.Linear(input_size, attention_size, bias=False)
        self.Q = nn.Linear(input_size, attention_size, bias=False)
        self.V = nn.Linear(input_size, attention_size, bias=False)
        self.output_layer = nn.Sequential(nn.Linear(attention_size, output_size, bias=False), nn.Dropout(dropout))


class LatentPSL(nn.Module):
    def __init__(self, input_size, num_psl):
        super().__init__()
        self.theta = nn.Parameter(torch.empty(num_psl, input_size))
        nn.init.xavier_uniform_(self.theta, gain=nn.init.calculate_gain('tanh'))
        self.out_norm = nn.Sequential(nn.Tanh(), nn.LayerNorm(input_size), nn.Dropout(0.3))


class EncoderVisual(nn.Module):
    def __init__(self, args, baseline=False):
        super().__init__()
        H = args.visual_hidden_size
        self.hidden_size = H
        self.p_drop = args.dropout
        self.linear_embed = nn.Linear(args.a_feature_size + args.m_feature_size, H)
        nn.init.xavier_normal_(self.linear_embed.weight)
        self.lstm = nn.LSTM(H, H, batch_first=True, bidirectional=True)
        self.layernorm_lstm = nn.LayerNorm(2 * H)
        self.baseline = baseline
        if not baseline:
            self.self_attention = SelfAttention(2 * H, 2 * H, H, args.dropout, True)
            self.layernorm_sa = nn.LayerNorm(H)
        else:
            self.out_try = nn.Linear(2 * H, H)
            nn.init.xavier_normal_(self.out_try.weight)


class EncoderVisualGraphTUN(nn.Module):
    def __init__(self, args, input_type='motion', use_embed=True, baseline=False):
        super().__init__()
        self.baseline = baseline
        self.has_obj = args.num_obj > 4
        if self.has_obj:
            self.obj_embed = nn.Linear(args.region_feature_size, args.region_projected_size)
            self.obj_norm = nn.Sequential(nn.Tanh(), nn.LayerNorm(args.region_projected_size))
        vin = args.m_feature_size if input_type == 'motion' else args.a_feature_size
        self.use_embed = use_embed
        if use_embed:
            self.visual_embed = nn.Linear(vin, args.visual_hidden_size)
        self.visual_norm = nn.Sequential(nn.Tanh(), nn.LayerNorm(args.visual_hidden_size))
        self.obj_visual_norm = nn.Sequential(nn.Tanh(), nn.LayerNorm(args.visual_hidden_size))
        self.v2l_layer = LatentPSL(args.visual_hidden_size, args.num_proposals)
        self.att_l2l_norm = nn.LayerNorm(args.visual_hidden_size)   # constructed, unused (layer.py:167)


class CapGnnEncoder(nn.Module):
    def __init__(self, args, baseline=False):
        super().__init__()
        self.a_feature_size = args.a_feature_size
        self.obj_encoder = EncoderVisualGraphTUN(args, 'object', baseline=baseline)
        self.motion_pre_encoder = EncoderVisual(args)
        self.motion_encoder = EncoderVisualGraphTUN(args, 'motion', use_embed=False, baseline=baseline)


class Decoder(nn.Module):
    def __init__(self, args, vocab, multi_modal=False, baseline=False):
        super().__init__()
        self.vocab = vocab
        self.vocab_size = len(vocab)
        self.word_size = args.word_size
        self.max_words = args.max_words
        self.beam_size = args.beam_size
        self.p_drop = args.dropout
        self.query_hidden_size = args.query_hidden_size
        self.decode_hidden_size = args.decode_hidden_size
        self.visual_hidden_size = args.visual_hidden_size
        self.multi_modal = multi_modal
        self.baseline = baseline
        H, W, Q, D = args.visual_hidden_size, args.word_size, args.query_hidden_size, args.decode_hidden_size
        self.word_embed = nn.Embedding(self.vocab_size, W)
        self.dataset = getattr(args, 'dataset', 'msvd')
        if getattr(args, 'use_glove', False):
            self.get_glove_embedding()
        q_in = H + W + D + (0 if baseline else H)
        self.query_lstm = nn.LSTMCell(q_in, Q)
        self.query_lstm_layernorm = nn.LayerNorm(Q)
        l_in = H + Q + (H if multi_modal else 0)
        self.lang_lstm = nn.LSTMCell(l_in, D)
        self.lang_lstm_layernorm = nn.LayerNorm(D)
        self.context_att = AttentionShare(H, Q, H)
        self.context_layernorm = nn.LayerNorm(D)                 # constructed, unused (layer.py:334)
        if multi_modal:
            self.context_att_2 = AttentionShare(H, Q, H)
        self.word_restore = nn.Linear(D, self.vocab_size)
        nn.init.xavier_normal_(self.word_restore.weight)

    def get_glove_embedding(self, data_dir='./data'):
        """models/layer.py:352-385 (`--use_glove`): the word embedding starts from `./data/{dataset}_glove.npy` (one row per
        vocabulary id); when that file is missing it is built from `./data/glove.42B.300d.txt` -- a word with a trailing comma
        is looked up without it, a word GloVe does not have gets N(0, 0.6) -- and written for the next run, as the reference
        does.  Neither file present: an error naming both (the reference dies on the open())"""
        import os
        import numpy as np
        npy = os.path.join(data_dir, '%s_glove.npy' % self.dataset)
        V, W = self.word_embed.num_embeddings, self.word_embed.embedding_dim
        if os.path.exists(npy):
            weight = np.load(npy)
        else:
            txt = os.path.join(data_dir, 'glove.42B.300d.txt')
            if not os.path.exists(txt):
                raise FileNotFoundError('use_glove: neither %s nor %s exists (models/layer.py:353-359)' % (npy, txt))
            wanted = {}
            for i, word in enumerate(self.vocab.idx2word):
                wanted.setdefault(word[:-1] if word.endswith(',') else word, []).append(i)
            weight = np.random.normal(scale=0.6, size=(V, W))
            with open(txt, 'rb') as f:
                for line in f:
                    parts = line.decode().split()
                    for i in wanted.get(parts[0], ()):
                        weight[i] = np.array(parts[1:]).astype(np.float64)
            np.save(npy, weight)
        if weight.shape != (V, W):
            raise ValueError('%s holds %s, the word embedding is (%d, %d)' % (npy, weight.shape, V, W))
        self.word_embed.load_state_dict({'weight': torch.from_numpy(np.asarray(weight))})

    def update_beam_size(self, beam_size):
        """models/layer.py:348-350"""
        self.beam_size = beam_size

    def __getstate__(self):
        d = self.__dict__.copy()
        d.pop('_owner', None)              # a weak reference to the owning model (set by the model): not part of the state
        return d

    def forward(self, cnn_feats, captions, max_words, teacher_forcing_ratio, cnn_feats_2=None, step_feats=None):
        """models/layer.py:394-447 called on its own (the models call the engine directly): teacher-forced / scheduled-sampling
        logits + attention weights, or greedy ids when `captions` is None, from given proposals; `step_feats` (B, G) replaces
        the means of the proposals as the global feature (layer.py:404-405; as there, a non-multi-modal decoder then attends
        over `cnn_feats` alone).  Forward only (the gradients of the path go through the model's autograd bridge); needs the
        owning model's kernel binding, so the decoder must belong to one of dlsg_amd's models.  With `captions` None and
        beam_size != 1 the reference's decoder runs its beam search itself (layer.py:449-460): so does this one
        (`beam.beam_search_from` with the given proposals), returning (ids of the best beam, [])."""
        from . import engine as E
        model = self._owner() if getattr(self, '_owner', None) is not None else None
        if model is None:
            raise RuntimeError('Decoder.forward needs the model that owns this decoder (its HIP binding and parameter arena)')
        infer = captions is None
        L = self.max_words if max_words is None else max_words
        model.flatten_parameters_()
        ops = model.ops
        model._gemm_policy(False)
        feats1 = cnn_feats.contiguous().float()
        feats2 = cnn_feats_2.contiguous().float() if cnn_feats_2 is not None else None
        sv = {'dec_gsrc': [feats1] + ([feats2] if feats2 is not None else [])}
        if step_feats is not None:
            sv['step_feats'] = step_feats.contiguous().float()
        if feats2 is not None and self.multi_modal:
            mems = [feats1, feats2]
        elif feats2 is not None and step_feats is None:
            mems = [torch.cat([feats1, feats2], 1)]          # layer.py:412-413
        else:
            mems = [feats1]
        if infer and self.beam_size != 1:
            # layer.py:449-460 (the search always runs self.max_words steps: BeamSearch(max_steps=max_words) is built in __init__)
            from .beam import beam_search_from, beam_finish
            return beam_finish(model, *beam_search_from(model, mems, sv, model.next_seed(), early_exit=True))[0], []
        coins = model._draw_coins(L, infer, teacher_forcing_ratio)
        with torch.no_grad():
            s = E.dec_fwd(ops, self, mems, sv, captions, L, coins, self.training and not infer, model.next_seed())
            if infer:
                return s['IDS'][1:].t().contiguous(), []
            B = feats1.shape[0]
            logits = torch.empty(B, L, self.vocab_size, dtype=torch.float32, device=feats1.device)
            ops.permute_tb(s['LOGITS'], logits)
            alpha = torch.empty(B, L, s['ALPHA'].shape[-1], dtype=torch.float32, device=feats1.device)
            ops.permute_tb(s['ALPHA'], alpha)
        return logits, [alpha[:, i].unsqueeze(2) for i in range(L)]

    def decode_tokens(self, tokens):
        """models/layer.py:464-477: ids -> words until <end>."""
        words = []
        end = self.vocab('<end>')
        for tok in tokens:
            tok = int(tok)
            if tok == end:
                break
            words.append(self.vocab.idx2word[tok])
        return ' '.join(words)
