"""TEST INFRASTRUCTURE ONLY -- oracle for SURVEY.md section 8(f) rank 1: the `DiscV2` critic and the WGAN-GP iteration.

A CPU restatement (plain torch, fp32) of
  * models/model.py:110-168  DiscV2.forward
  * models/layer.py:661-715  PSLScore2
  * models/sublayer.py:63-82 SelfAttention (masked form), :107-119 ResBlock, :189-198 LatentPSL, :292-306 JointEmbedVideoModel2
  * run_gun.py:339-398       RunGAN.train_disc (num_D critic steps, gradient penalty with create_graph)
  * run_gun.py:153-234       one iteration: no-grad generator forward, critic steps, generator step with
                             total_loss = cap_loss + gan_lambda * (-D(tokens).mean())
Only tests/, __graft_entry__.smoke() and bench.py's baseline legs may import it; the product (dlsg_amd/gan.py) never does.
PINNED by tests/golden/gan_*.npz, which tests/golden/make_goldens_r2.py produced by running the reference's own DiscV2 /
CapGnnModel (tests/test_oracle_golden.py::test_gan_*).

The module tree mirrors the reference's attribute names because `state_dict` keys are the checkpoint contract
(`model_d_state_dict`, run_gun.py:306); the forward is a functional restatement.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import torch_ref as R


class _ResBlockP(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.res_block = nn.Sequential(nn.ReLU(True), nn.Conv1d(dim, dim, 3, padding=1))


class _JointEmbedP(nn.Module):
    def __init__(self, h):
        super().__init__()
        self.classify = nn.Linear(h, 1)
        self.visual_embed = nn.Sequential(nn.Linear(h, h), nn.Tanh())
        self.sent_embed = nn.Sequential(nn.Linear(h, h), nn.Tanh())


class _PSLScore2P(nn.Module):
    def __init__(self, num_psl, num_top):
        super().__init__()
        self.psl_scorer = _JointEmbedP(512)
        self.psl_embed = nn.Sequential(nn.Linear(1024, 512), nn.Tanh(), nn.LayerNorm(512))
        self.psl_norm = nn.Sequential(nn.Tanh(), nn.LayerNorm(512), nn.Dropout(0.3))
        self.att_norm = nn.Sequential(nn.Linear(512, 512), nn.Tanh(), nn.LayerNorm(512))
        self.num_top = num_top
        self.select = num_psl > num_top


class DiscV2Ref(nn.Module):
    """models/model.py:110-134 (parameters) + :143-166 (forward)."""

    def __init__(self, opt, vocab_size):
        super().__init__()
        self.dim = 512
        self.num_top = opt.num_topk
        self.seq_len = opt.max_words
        self.num_psl = opt.num_proposals
        self.block = nn.Sequential(_ResBlockP(self.dim))
        self.conv1d = nn.Conv1d(vocab_size, self.dim, 1)
        self.lstm = nn.LSTM(512, 512, batch_first=True, bidirectional=False)
        self.layer_norm = nn.LayerNorm(512)
        self.lstm_drop = nn.Dropout(0.3)
        self.att = R.SelfAttentionP(512, 512, 512, 0.3)
        self.att_norm = nn.Sequential(nn.Tanh(), nn.LayerNorm(512))
        self.motion_psl_score = _PSLScore2P(opt.num_proposals, self.num_top)
        self.obj_psl_score = _PSLScore2P(opt.num_proposals, self.num_top)
        self.text_sum = R.LatentPSLP(512, 1)
        self.fusion = nn.Parameter(torch.empty(2, 512))
        nn.init.xavier_uniform_(self.fusion, gain=nn.init.calculate_gain('tanh'))

    def forward(self, inputs, obj_proposals, motion_proposals, att_mask=None, alpha_all=None):
        return disc_forward(self, inputs, obj_proposals, motion_proposals, att_mask, alpha_all)


def _drop(x, p, training):
    return F.dropout(x, p, training) if training and p > 0 else x


def _ln(x, mod):
    return F.layer_norm(x, mod.normalized_shape, mod.weight, mod.bias, mod.eps)


def psl_score2(m, psl, psl_alpha, att_out, seq_mask, training):
    """models/layer.py:690-715"""
    bs = psl.size(0)
    e = _ln(torch.tanh(m.psl_embed[0](psl)), m.psl_embed[2])
    if m.select:
        idx = torch.topk(psl_alpha.sum(dim=1), m.num_top, -1)[1]
        e = torch.gather(e, 1, idx.unsqueeze(-1).expand(bs, m.num_top, e.size(-1)))
    a = _ln(torch.tanh(m.att_norm[0](att_out)), m.att_norm[2])                           # (B,L,512)
    adj = torch.matmul(a, e.transpose(-1, -2)) / math.sqrt(512)                          # (B,L,k)
    adj = F.softmax(adj, dim=1)                                                          # over the words
    adj = torch.where(seq_mask > 0, adj, torch.zeros_like(adj))                          # mask AFTER the softmax (:703-704)
    adj_alpha = adj.sum(dim=1)                                                           # (B,k)
    agg = torch.matmul(a.transpose(-1, -2), adj).transpose(-1, -2)                       # (B,k,512)
    agg = _drop(_ln(torch.tanh(agg), m.psl_norm[1]), 0.3, training)
    sc = m.psl_scorer
    score = sc.classify(torch.tanh(sc.visual_embed[0](e)) * torch.tanh(sc.sent_embed[0](agg))).squeeze()   # (B,k)
    score = (score * adj_alpha).sum(dim=-1) / adj_alpha.sum(dim=-1)
    return score.mean(dim=-1)                                                            # mean over a 0-d/1-d tail, as :714


def disc_forward(m, inputs, obj, mot, att_mask, alpha_all):
    """models/model.py:143-166.  inputs (B,L,V) one-hot captions or raw logits."""
    training = m.training
    x = m.conv1d(inputs.transpose(1, 2))                                                 # (B,512,L)
    conv = m.block[0].res_block[1]
    # ResBlock (sublayer.py:110-119): its ReLU is IN PLACE (nn.ReLU(True)), so the skip connection carries relu(x), not x
    x = F.relu(x)
    x = x + 0.3 * conv(x)
    h, _ = m.lstm(x.transpose(1, 2))
    h = _drop(_ln(h, m.layer_norm), 0.3, training)
    att = R.self_attention(m.att, h, att_mask, training, get_pe=False)          # DiscV2 builds it without positional encoding
    att = _ln(torch.tanh(att), m.att_norm[1])
    P = m.num_psl
    word_mask = att_mask[:, 0, :].unsqueeze(2)
    alpha = alpha_all * word_mask.repeat(1, 1, 2 * P)
    mask_k = word_mask.repeat(1, 1, m.num_top)
    so = psl_score2(m.obj_psl_score, obj, alpha[:, :, :P], att, mask_k, training)
    sm = psl_score2(m.motion_psl_score, mot, alpha[:, :, -P:], att, mask_k, training)
    sent = R.latent_psl(m.text_sum, att, training).squeeze()                              # (B,512)
    fus = F.softmax(torch.matmul(sent, m.fusion.t()), dim=-1)
    return so * fus[:, 0] + sm * fus[:, 1]


def to_onehot(seq, vocab_size):
    """run_gun.py:447-451"""
    return F.one_hot(seq, vocab_size).to(torch.float32)


def attention_mask(captions):
    """run_gun.py:164-166: outer product of the non-pad word mask"""
    seq = (captions > 0).to(torch.float32)
    return seq.unsqueeze(2) * seq.unsqueeze(1)


def critic_losses(D, r_caption, f_caption, obj, mot, att_mask, alpha, eps_gp):
    """One pass of run_gun.py:345-375.  eps_gp (B,1,1) in [0,1).  Returns (loss_D, r_loss, f_loss, gradient_penalty, logits)."""
    r_logit = D(r_caption, obj, mot, att_mask, alpha)
    f_logit = D(f_caption, obj, mot, att_mask, alpha)
    eps_gp = eps_gp.clone().requires_grad_(True)
    mixed = r_caption.detach() * eps_gp + f_caption.detach() * (1 - eps_gp)
    m_logit = D(mixed, obj, mot, att_mask, alpha)
    g = torch.autograd.grad(inputs=mixed, outputs=m_logit, grad_outputs=torch.ones_like(m_logit), create_graph=True,
                            retain_graph=True)[0]
    gn = g.contiguous().view(len(g), -1).norm(2, dim=1)
    gp = ((gn - 1) * (gn - 1)).mean()
    r_loss, f_loss = r_logit.mean(), f_logit.mean()
    return f_loss - r_loss + 10 * gp, r_loss, f_loss, gp, (r_logit, f_logit, m_logit)


def train_disc(D, opt_D, r_caption, f_caption, obj, mot, att_mask, alpha, num_D, eps_list):
    """run_gun.py:339-381.  eps_list: num_D tensors (B,1,1) standing in for torch.rand.  Returns (mean loss_D, mean wasserstein)."""
    mean_loss, mean_w = 0.0, 0.0
    for k in range(num_D):
        opt_D.zero_grad()
        loss_D, r_loss, f_loss, gp, _ = critic_losses(D, r_caption, f_caption, obj, mot, att_mask, alpha, eps_list[k])
        mean_loss += loss_D.item() / num_D
        mean_w += (r_loss.item() - f_loss.item()) / num_D
        loss_D.backward()
        opt_D.step()
    return mean_loss, mean_w


def gan_iteration(G, D, opt_G, opt_D, frames, regions, captions, cap_lens, tf_ratio, gan_lambda, num_D, eps_list, max_len=26):
    """run_gun.py:153-234 with use_visual_gan: returns dict(cap_loss, loss_G, total_loss, loss_D, wasserstein)."""
    captions = captions[:, :max_len]
    V = D.conv1d.in_channels
    att_mask = attention_mask(captions)
    f_caption, obj, mot, alpha = G(frames, regions, captions, max_len, tf_ratio)           # run_gun.py:167
    f_caption, obj, mot, alpha = f_caption.detach(), obj.detach(), mot.detach(), alpha.detach()
    r_caption = to_onehot(captions, V)
    loss_D, wass = train_disc(D, opt_D, r_caption, f_caption, obj, mot, att_mask, alpha, num_D, eps_list)
    opt_G.zero_grad()
    outputs, obj, mot, alpha = G(frames, regions, captions, max_len, tf_ratio)             # run_gun.py:183
    cap_loss = R.ragged_ce(outputs, captions, cap_lens)
    f_logit = D(outputs, obj.detach(), mot.detach(), att_mask=att_mask, alpha_all=alpha.detach())
    loss_G = -f_logit.mean()
    total = cap_loss + loss_G * gan_lambda
    total.backward()
    opt_G.step()
    return dict(cap_loss=cap_loss.item(), loss_G=loss_G.item(), total_loss=total.item(), loss_D=loss_D, wasserstein=wass)


class GANLambdaHandlerRef(object):
    """utils/utils.py:196-265: adaptive weight of the generator's GAN loss, driven by the running caption loss."""

    def __init__(self, total_step, gan_lambda, cap_list=None):
        self.cap_list = list(cap_list) if cap_list is not None else []
        self.current_step = 0
        self.total_step = total_step
        self.counter = 500
        self.current_schedule_step = 0
        self.start_gan_lambda = gan_lambda
        self.low_gan_lambda = 0.006
        self.increase_schedule = self._schedule(1.5, 2.5)
        self.decrease_schedule = self._schedule(0.5, 1.5)
        self.current_lambda = gan_lambda
        self.state = 0                      # 0 stable, 1 decrease, 2 increase

    def _schedule(self, lo, hi):
        base = (self.start_gan_lambda - self.low_gan_lambda) / 2
        x = np.arange(int(self.counter * hi))[int(self.counter * lo):]
        return (np.sin(2 * np.pi * 0.5 * x / self.counter) * base + base + self.low_gan_lambda).tolist()

    def update_gan_lambda(self, epoch, i, cap_loss):
        self.current_step = i - 1 + epoch * self.total_step
        self.cap_list.append(cap_loss)
        width = 200
        if len(self.cap_list) > width:
            self.cap_list = self.cap_list[-width:]
            if self.state == 0:
                first = np.mean(np.array(self.cap_list[:width // 2]))
                last = np.mean(np.array(self.cap_list[width // 2:]))
                if last > first * 1.04:
                    self.state = 1
            elif self.current_schedule_step == self.counter - 1:
                self.current_schedule_step = 0
                self.state = 0

    def get_current_lambda(self):
        if self.state == 1:
            self.current_lambda = self.decrease_schedule[self.current_schedule_step]
            self.current_schedule_step += 1
        elif self.state == 2:
            self.current_lambda = self.increase_schedule[self.current_schedule_step]
            self.current_schedule_step += 1
        return self.current_lambda
