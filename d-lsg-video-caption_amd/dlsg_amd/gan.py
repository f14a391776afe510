"""the `DiscV2` critic and the WGAN-GP iteration `train_debug.py` really trains
(models/model.py:110-168, models/layer.py:661-715, run_gun.py:153-234,339-398).

Split of work on the GPU:
  * generator side -- both forwards of an iteration (the no-grad one of run_gun.py:167 and the trained one of :183), the
    ragged CrossEntropy, the whole backward and Adam -- run on this repo's HIP kernels through `Trainer`; the critic's
    gradient w.r.t. the logits enters the hand-scheduled backward as an extra d(logits) term (`Trainer.step(extra_dlogits=)`).
    The reference detaches the proposals and attention weights before they reach the critic (run_gun.py:214-217), so
    d(logits) is the only path from the GAN loss into the generator.
  * critic side -- PyTorch-ROCm eager autograd (it needs a double backward for the gradient penalty, run_gun.py:362-371;
    SURVEY.md 8f: "keep in PyTorch-ROCm eager first"), restructured so that nothing of size (B, L, V) is materialised:
      - Conv1d(V -> 512, k = 1) is a linear map over the vocabulary axis: real captions go through an embedding GATHER of
        its weight columns instead of a one-hot (B, L, V) x (V, 512) product (run_gun.py:447-451 builds the one-hot);
      - by the same linearity the gradient-penalty sample eps*real + (1-eps)*fake is mixed AFTER the projection, so the
        three critic forwards of a step (real, fake, mixed) share ONE logit projection and run as one 3B-row batch;
      - |d mixed_logit / d mixed_captions|^2 = sum_l g_l (W W^T) g_l^T with g = d mixed_logit / d(projection): a 512 x 512
        Gram matrix replaces the (B, L, V) gradient tensor, and stays differentiable for the second backward;
      - the LSTM is unrolled over plain matmuls (MIOpen's fused RNN has no double backward; the reference switches cuDNN off
        for the same reason, train_debug.py:53).
    All of it is exact up to fp32 reassociation; tests compare against the reference's own numbers (tests/golden/gan_*.npz).

`DiscV2.state_dict()` has the reference's keys and shapes (checkpoint key `model_d_state_dict`, run_gun.py:306).
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .modules import LatentPSL, SelfAttention

WIDTH = 512          # DiscV2.dim and every internal width (models/model.py:113, layer.py:665-683)
PSL_WIDTH = 1024     # proposals enter through Linear(1024, 512) (layer.py:666)


class _Residual(nn.Module):
    """parameter holder with the reference's key `res_block.1.{weight,bias}` (sublayer.py:107-115)"""

    def __init__(self, dim):
        super().__init__()
        self.res_block = nn.Sequential(nn.ReLU(True), nn.Conv1d(dim, dim, 3, padding=1))


class _PairScorer(nn.Module):
    """JointEmbedVideoModel2 parameters (sublayer.py:292-303)"""

    def __init__(self, h):
        super().__init__()
        self.classify = nn.Linear(h, 1)
        self.visual_embed = nn.Sequential(nn.Linear(h, h), nn.Tanh())
        self.sent_embed = nn.Sequential(nn.Linear(h, h), nn.Tanh())


class _ProposalScore(nn.Module):
    """PSLScore2 parameters (layer.py:661-688)"""

    def __init__(self, num_psl, num_top):
        super().__init__()
        self.psl_scorer = _PairScorer(WIDTH)
        self.psl_embed = nn.Sequential(nn.Linear(PSL_WIDTH, WIDTH), nn.Tanh(), nn.LayerNorm(WIDTH))
        self.psl_norm = nn.Sequential(nn.Tanh(), nn.LayerNorm(WIDTH), nn.Dropout(0.3))
        self.att_norm = nn.Sequential(nn.Linear(WIDTH, WIDTH), nn.Tanh(), nn.LayerNorm(WIDTH))
        self.num_top = num_top
        self.select = num_psl > num_top


def _tanh_ln(x, ln, ops=None, pre_tanh=True):
    """LayerNorm(tanh(x)) (or LayerNorm(x)); with a kernel interface and a 64..1024-wide row, the fused three-level op"""
    N = x.shape[-1]
    if ops is not None and N % 64 == 0 and N <= 1024 and (x.dtype == torch.float32 or getattr(ops, 'name', '') != 'hip'):
        return _TanhLN.apply(ops, x, ln.weight, ln.bias, ln.eps, pre_tanh)
    return F.layer_norm(torch.tanh(x) if pre_tanh else x, ln.normalized_shape, ln.weight, ln.bias, ln.eps)


class _Taps(torch.autograd.Function):
    """x (n, L, C) -> (n, L, 3C) = [x[t-1] | x[t] | x[t+1]] (adjoint=False) or the transposed map (adjoint=True): one launch each
    (csrc/critic.hip conv_taps_kernel), each the other's backward -- instead of pad + three slices + cat and, per backward level,
    three zero-filled (n, L + 2, C) slice gradients with their copies and adds."""

    @staticmethod
    def forward(ctx, ops, x, adjoint):
        x = x.contiguous()
        C_ = x.shape[2] // 3 if adjoint else x.shape[2]
        y = x.new_empty(x.shape[0], x.shape[1], C_ if adjoint else 3 * C_)
        ops.conv_taps(x, y, adjoint)
        ctx.ops, ctx.adjoint = ops, adjoint
        return y

    @staticmethod
    def backward(ctx, dy):
        return None, _Taps.apply(ctx.ops, dy, not ctx.adjoint), None


def _tanh_ln_stacked(x, gamma, beta, eps, ops):
    """x (G, R, N), gamma / beta (G, N): G same-shape LayerNorm(tanh(x[g])) in one launch per differentiation level"""
    return _TanhLN.apply(ops, x, gamma, beta, eps, True)


def _linear_stacked(ops, x, W, b=None):
    """x (G, R, K) @ W (G, N, K)^T + b (G, N): G same-shape linear layers as one batched product"""
    g = _gemm_ops(ops, GEMM_NT, x, W)
    if g is not None:
        y = _Gemm.apply(g, GEMM_NT, x, W, None, 1.0)
        return y if b is None else y + b.unsqueeze(1)
    if b is None:
        return torch.bmm(x, W.transpose(1, 2))
    return torch.baddbmm(b.unsqueeze(1), x, W.transpose(1, 2))


def _dropout(x, p, on):
    return F.dropout(x, p, True) if on and p > 0 else x


_HIP = []


def _hip_ops():
    if not _HIP:
        from .hip import HipOps
        _HIP.append(HipOps())
    return _HIP[0]


GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2          # include/dlsg.h: C = A B^T | A B | A^T B (row-major operands)


def _row_major(t):
    """an operand csrc/gemm.hip can address: unit stride along the last axis (any row / batch stride)"""
    return t if (t.stride(-1) == 1 or t.size(-1) == 1) else t.contiguous()


def _tn_chunks(M, N, K):
    """A^T B with a deep contraction and a small output (the weight gradients over all 4 992 caption rows: 512 x 512 is 64
    tiles on 256 CUs): the K rows go to this many groups of one launch, each writing its own slab"""
    if K < 2048:
        return 1
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    return max(1, min(16, 768 // tiles, K // 256))


def _gemm_tn_split(ops, A, B, C, alpha):
    K = A.shape[0]
    ks = _tn_chunks(C.shape[0], C.shape[1], K)
    step = ((K + ks - 1) // ks + 31) // 32 * 32
    bounds = [(k, min(K, k + step)) for k in range(0, K, step)]
    slabs = C.new_empty(len(bounds), C.shape[0], C.shape[1])
    ops.gemm(GEMM_TN, [(A[k0:k1], B[k0:k1], slabs[i]) for i, (k0, k1) in enumerate(bounds)], alpha=alpha)
    ops.slab_reduce(slabs, C)


def _product(ops, mode, A, B, C, alpha=1.0, bias=None):
    """C = alpha op(A) op(B) (+ bias) on the kernel that fits the shape: narrow (one side <= 32), deep TN split, or the tiled GEMM"""
    K = A.shape[-2] if mode == GEMM_TN else A.shape[-1]
    nb = C.shape[0] if C.dim() == 3 else 1
    kind = ops.gemm_narrow_kind(mode, C.shape[-2], C.shape[-1], K, nb) if hasattr(ops, 'gemm_narrow_kind') else 0
    if kind and not (kind == 3 and bias is not None):
        ops.gemm_narrow(mode, A, B, C, alpha, bias)
    elif mode == GEMM_TN and A.dim() == 2 and bias is None and _tn_chunks(C.shape[0], C.shape[1], K) > 1:
        _gemm_tn_split(ops, A, B, C, alpha)
    else:
        ops.gemm(mode, [(A, B, C)], alpha=alpha, bias=bias)


class _Gemm(torch.autograd.Function):
    """C = alpha * op(A) op(B) (+ bias) on this repo's GEMM (csrc/gemm.hip, exact fp32 MFMA) instead of rocBLAS: 2-D operands, or
    3-D ones as a batch.  The two gradient products are `_Gemm` nodes again (the three layouts are closed under
    differentiation), so the gradient penalty's double backward runs on the same kernel."""

    @staticmethod
    def forward(ctx, ops, mode, A, B, bias, alpha):
        A, B = _row_major(A), _row_major(B)
        if mode == GEMM_NT:
            M, N = A.shape[-2], B.shape[-2]
        elif mode == GEMM_NN:
            M, N = A.shape[-2], B.shape[-1]
        else:
            M, N = A.shape[-1], B.shape[-1]
        C = A.new_empty(*A.shape[:-2], M, N)
        _product(ops, mode, A, B, C, alpha, None if bias is None else bias.contiguous())
        ctx.ops, ctx.mode, ctx.alpha, ctx.has_bias = ops, mode, alpha, bias is not None
        ctx.save_for_backward(A, B)
        return C

    @staticmethod
    def backward(ctx, dC):
        A, B = ctx.saved_tensors
        ops, mode, al = ctx.ops, ctx.mode, ctx.alpha
        dA = dB = db = None
        need_a, need_b = ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        dC = _row_major(dC)
        if mode == GEMM_NT:          # C = A B^T
            if need_a:
                dA = _mm(ops, GEMM_NN, dC, B, al)
            if need_b:
                dB = _mm(ops, GEMM_TN, dC, A, al)
        elif mode == GEMM_NN:        # C = A B
            if need_a:
                dA = _mm(ops, GEMM_NT, dC, B, al)
            if need_b:
                dB = _mm(ops, GEMM_TN, A, dC, al)
        else:                        # C = A^T B
            if need_a:
                dA = _mm(ops, GEMM_NT, B, dC, al)
            if need_b:
                dB = _mm(ops, GEMM_NN, A, dC, al)
        if ctx.has_bias and ctx.needs_input_grad[4]:
            db = dC.reshape(-1, dC.shape[-1]).sum(0)
        return None, None, dA, dB, db, None


def _gemm_ops(ops, mode, A, B):
    """which backend multiplies: the kernel interface (returned) or ATen / rocBLAS (None).  DLSG_CRITIC_GEMM = rocblas (default) |
    dlsg (every product of a critic update on csrc/gemm.hip + csrc/gemm_narrow.hip: no vendor GEMM on the path; parity-tested,
    tests/test_gpu_gan.py).  Measured on an MI355X at batch 64 (tools/critic_gemm_census.py, profiles/r03k_critic_gemm_census.txt):
    the 175 products of an update take 4.1 ms on rocBLAS and 5.4 ms on this repo's kernels, which are tuned for the generator's
    shapes (26 624-row projections, 64-row recurrences) -- they win the tall 4 992 x 512 x 512 products (36 vs 43 us) and lose the
    deep 512 x 512 x 4 992 weight gradients (44 vs 35 us), the 576-row ones (16 vs 8 us: a 64 x 64 fp32-MFMA tile needs 7 us for
    K = 512) and most 26 x 26 / 26 x 3 batched ones; 5 critic updates: 55 ms against 44.  A per-shape mix of the two measured
    no better than rocBLAS alone (44.2 vs 44.3 ms), so the default stays one backend.
    A kernel interface that is not the HIP library (the tests' emulation) always takes the products, so the CPU tests cover the
    whole three-level algebra of `_Gemm`."""
    if ops is None or not hasattr(ops, 'gemm'):
        return None
    if getattr(ops, 'name', '') != 'hip':
        return ops
    if A.dtype != torch.float32 or B.dtype != torch.float32:
        return None
    return ops if os.environ.get('DLSG_CRITIC_GEMM', 'rocblas') == 'dlsg' else None


def _mm(ops, mode, A, B, alpha=1.0, bias=None):
    """alpha * op(A) op(B) (+ bias): both 2-D or both 3-D (a batch), on the backend `_gemm_ops` picks for the shape"""
    g = _gemm_ops(ops, mode, A, B)
    if g is not None:
        return _Gemm.apply(g, mode, A, B, bias, alpha)
    if mode == GEMM_NT:
        if bias is not None and alpha == 1.0 and A.dim() == 2:
            return F.linear(A, B, bias)
        r = A @ B.transpose(-1, -2)
    elif mode == GEMM_NN:
        r = A @ B
    else:
        r = A.transpose(-1, -2) @ B
    r = r if alpha == 1.0 else r * alpha
    return r if bias is None else r + bias


def _linear(ops, x, W, b=None):
    """F.linear over the last axis, the bias added in the product's epilogue"""
    y = _mm(ops, GEMM_NT, x.reshape(-1, x.shape[-1]), W, 1.0, b)
    return y.view(*x.shape[:-1], W.shape[0])


def _seq_kernels(ops, L, n, H):
    """whole-sequence launches where the op set has them and the shape fits (n <= 256, H in {64, 512}, enough CUs)"""
    f = getattr(ops, 'lstm_seq_supported', None)
    return f is not None and f(L, n, H)


class _LstmSeq(torch.autograd.Function):
    """The critic's whole LSTM layer as ONE autograd node per differentiation level (zero initial state, gates i,f,g,o):
        a_t = xin_t + h_{t-1} W^T,  (h_t, c_t) = cell(a_t, c_{t-1})        xin (L, n, 4H) already holds x W_ih^T + b
    returns (Hs, As, Cs); As and Cs are outputs so that the backward's backward can hand its gradients w.r.t. the saved
    pre-activations and cell states back to this node.  Every level is ONE persistent launch over all word steps
    (csrc/critic_lstm.hip) where the shape fits, else one recurrent product plus one cell kernel per step (csrc/critic.hip);
    weight gradients are single products over all steps -- no per-step gradient accumulation."""

    @staticmethod
    def forward(ctx, ops, xin, W):
        L, n, G = xin.shape
        H = G // 4
        Hs, Cs = xin.new_empty(L, n, H), xin.new_empty(L, n, H)
        if _seq_kernels(ops, L, n, H):                  # all L steps in one launch (csrc/critic_lstm.hip)
            As = torch.empty_like(xin, memory_format=torch.contiguous_format)
            ops.lstm_seq_fwd(xin.contiguous(), W.contiguous(), As, Hs, Cs)
        else:
            As = xin.clone()
            Wt = W.t()
            for t in range(L):
                if t:
                    As[t].addmm_(Hs[t - 1], Wt)
                ops.lstm_cell_fwd(As[t], Cs[t - 1] if t else None, Hs[t], Cs[t])
        ctx.ops = ops
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(As, Cs, Hs, W)
        return Hs, As, Cs

    @staticmethod
    def backward(ctx, dHs, dAs, dCs):
        As, Cs, Hs, W = ctx.saved_tensors
        if dHs is None:
            dHs = torch.zeros_like(Hs)
        dxin, dW = _LstmSeqBwd.apply(ctx.ops, As, Cs, Hs, W, dHs.contiguous(), None if dAs is None else dAs.contiguous(),
                                     None if dCs is None else dCs.contiguous())
        return None, dxin, dW


class _LstmSeqBwd(torch.autograd.Function):
    """Backward through time of `_LstmSeq` with gradients injected on h (dHs), on the pre-activations (dAs) and on the cell
    states (dCs), itself differentiable:
        dh_t = dHs_t + DA_{t+1} W,  dc_t = s_t + dCs_t,  (da_t, s_{t-1}) = cell'(a_t, c_{t-1}; dh_t, dc_t),  DA_t = da_t + dAs_t
        dxin = DA,  dW = sum_{t>=1} DA_t^T h_{t-1}
    Its backward (cotangents Uxin on DA, UW on dW) runs forward in time:
        ubar_t = Uxin_t + h_{t-1} UW^T + gdh_{t-1} W^T
        (ga_t, gc_{t-1}, gdh_t, gdc_t) = cell''(a_t, c_{t-1}, dh_t, dc_t; ubar_t, gdc_{t-1})
    and returns gAs = ga, gCs_{t-1} = gc_{t-1}, gHs_{t-1} = DA_t UW, gW = sum DA_t^T gdh_{t-1}, g(dHs) = gdh, g(dAs) = ubar,
    g(dCs) = gdc."""

    @staticmethod
    def forward(ctx, ops, As, Cs, Hs, W, dHs, dAs, dCs):
        L, n, G = As.shape
        H = G // 4
        DA = torch.empty_like(As)
        DH, DC = torch.empty_like(Hs), torch.empty_like(Hs)          # the summed dh_t / dc_t each step was differentiated at
        if _seq_kernels(ops, L, n, H):
            ops.lstm_seq_bwd(As, Cs, W.contiguous(), dHs, dAs, dCs, DA, DH, DC)
        else:
            s_buf = [torch.empty_like(Hs[0]), torch.empty_like(Hs[0])]
            r = torch.empty_like(Hs[0])
            for t in range(L - 1, -1, -1):
                last = t == L - 1
                ops.lstm_cell_bwd_seq(As[t], Cs[t - 1] if t else None, dHs[t], None if last else r, None if last else s_buf[(t + 1) & 1],
                                      None if dCs is None else dCs[t], None if dAs is None else dAs[t], DA[t], s_buf[t & 1], DH[t], DC[t])
                if t:
                    torch.mm(DA[t], W, out=r)
        dW = DA[1:].reshape(-1, G).t() @ Hs[:-1].reshape(-1, H) if L > 1 else torch.zeros_like(W)
        ctx.ops = ops
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(As, Cs, Hs, W, DA, DH, DC)
        ctx.has = (dAs is not None, dCs is not None)
        return DA, dW

    @staticmethod
    def backward(ctx, Uxin, UW):
        As, Cs, Hs, W, DA, DH, DC = ctx.saved_tensors
        L, n, G = As.shape
        H = G // 4
        ops = ctx.ops
        Ubar = torch.zeros_like(As) if Uxin is None else Uxin.clone(memory_format=torch.contiguous_format)
        if UW is not None and L > 1:
            Ubar[1:].view(-1, G).addmm_(Hs[:-1].reshape(-1, H), UW.t())
        gA, gC, gDH, gDC = torch.empty_like(As), torch.zeros_like(Cs), torch.empty_like(Hs), torch.empty_like(Hs)
        if _seq_kernels(ops, L, n, H):
            ops.lstm_seq_bwd2(As, Cs, W.contiguous(), DH, DC, Ubar, gA, gC, gDH, gDC)
        else:
            Wt = W.t()
            for t in range(L):
                if t:
                    Ubar[t].addmm_(gDH[t - 1], Wt)
                ops.lstm_cell_bwd2(As[t], Cs[t - 1] if t else None, DH[t], DC[t], Ubar[t], gDC[t - 1] if t else None, gA[t],
                                   gC[t - 1] if t else None, gDH[t], gDC[t])
        gW = gHs = None
        if L > 1:
            gW = DA[1:].reshape(-1, G).t() @ gDH[:-1].reshape(-1, H)
            if UW is not None:
                gHs = torch.zeros_like(Hs)
                torch.mm(DA[1:].reshape(-1, G), UW, out=gHs[:-1].view(-1, H))
        return (None, gA, gC, gHs, gW, gDH, Ubar if ctx.has[0] else None, gDC if ctx.has[1] else None)


class _Softmax(torch.autograd.Function):
    """softmax over `dim` of a dense tensor, one launch per differentiation level (31 ATen launches per instance otherwise);
    its backward takes the OUTPUT y as a tracked input, so the second-order term reaches x through this node again."""

    @staticmethod
    def forward(ctx, ops, x, dim):
        x = x.contiguous()
        dim = dim % x.dim()
        outer = int(math.prod(x.shape[:dim]))
        inner = int(math.prod(x.shape[dim + 1:]))
        y = torch.empty_like(x)
        ops.softmax_fwd(x, y, outer, x.shape[dim], inner)
        ctx.ops, ctx.geom = ops, (outer, x.shape[dim], inner)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, = ctx.saved_tensors
        return None, _SoftmaxBwd.apply(ctx.ops, y, dy.contiguous(), ctx.geom), None


class _SoftmaxBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ops, y, dy, geom):
        dx = torch.empty_like(y)
        ops.softmax_bwd(y, dy, dx, *geom)
        ctx.ops, ctx.geom = ops, geom
        ctx.save_for_backward(y, dy)
        return dx

    @staticmethod
    def backward(ctx, u):
        y, dy = ctx.saved_tensors
        gy, gdy = torch.empty_like(y), torch.empty_like(y)
        ctx.ops.softmax_bwd2(y, dy, u.contiguous(), gy, gdy, *ctx.geom)
        return None, gy, gdy, None


def _softmax(x, dim, ops=None):
    if ops is not None and (x.dtype == torch.float32 or getattr(ops, 'name', '') != 'hip'):
        return _Softmax.apply(ops, x, dim)
    return torch.softmax(x, dim=dim)


class _TanhLN(torch.autograd.Function):
    """y = LayerNorm(tanh(x) or x): one HIP launch per differentiation level (csrc/critic.hip) instead of ~85 ATen launches
    per instance across forward, backward and the backward of the backward."""

    @staticmethod
    def forward(ctx, ops, x, gamma, beta, eps, pre_tanh):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        y = torch.empty_like(x2)
        ops.tanh_ln_fwd(x2, gamma.contiguous(), beta.contiguous(), y, eps, pre_tanh)
        ctx.ops, ctx.eps, ctx.pre_tanh = ops, eps, pre_tanh
        ctx.save_for_backward(x, gamma)          # the INPUT itself: a reshaped copy made in here has no history, and the
        return y.view(x.shape)                   # backward of the backward must reach x through it

    @staticmethod
    def backward(ctx, dy):
        x, gamma = ctx.saved_tensors
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        dx, dg, db = _TanhLNBwd.apply(ctx.ops, x2, gamma, dy.reshape(x2.shape).contiguous(), ctx.eps, ctx.pre_tanh)
        return None, dx.view(x.shape), dg, db, None, None


class _TanhLNBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ops, x2, gamma, dy, eps, pre_tanh):
        dx, dg, db = torch.empty_like(x2), torch.empty_like(gamma), torch.empty_like(gamma)
        g = gamma.contiguous()
        ops.tanh_ln_bwd(x2, g, dy, dx, dg, db, eps, pre_tanh)
        ctx.ops, ctx.eps, ctx.pre_tanh = ops, eps, pre_tanh
        ctx.save_for_backward(x2, g, dy)
        return dx, dg, db

    @staticmethod
    def backward(ctx, U, vg, vb):
        x2, g, dy = ctx.saved_tensors
        gx, gg, gdy = torch.empty_like(x2), torch.empty_like(g), torch.empty_like(dy)
        ctx.ops.tanh_ln_bwd2(x2, g, dy, U.contiguous(), vg.contiguous(), vb.contiguous(), gx, gg, gdy, ctx.eps, ctx.pre_tanh)
        return None, gx, gg, gdy, None, None


class DiscV2(nn.Module):
    """Critic of the visual GAN.  forward(inputs (B,L,V), obj (B,P,1024), mot (B,P,1024), att_mask (B,L,L), alpha_all (B,L,2P))
    -> (B,) scores, the reference's call (models/model.py:143); `score_projected` is the entry the trainer uses."""

    def __init__(self, opt, vocab_size):
        super().__init__()
        if opt.visual_hidden_size != PSL_WIDTH:
            raise ValueError('DiscV2 takes %d-wide proposals (models/layer.py:666); visual_hidden_size = %d'
                             % (PSL_WIDTH, opt.visual_hidden_size))
        self.dim = WIDTH
        self.num_top = opt.num_topk
        self.seq_len = opt.max_words
        self.num_psl = opt.num_proposals
        self.block = nn.Sequential(_Residual(WIDTH))
        self.conv1d = nn.Conv1d(vocab_size, WIDTH, 1)
        self.lstm = nn.LSTM(WIDTH, WIDTH, batch_first=True, bidirectional=False)       # parameters only: see _lstm
        self.layer_norm = nn.LayerNorm(WIDTH)
        self.att = SelfAttention(WIDTH, WIDTH, WIDTH, 0.3)
        self.att_norm = nn.Sequential(nn.Tanh(), nn.LayerNorm(WIDTH))
        self.motion_psl_score = _ProposalScore(opt.num_proposals, self.num_top)
        self.obj_psl_score = _ProposalScore(opt.num_proposals, self.num_top)
        self.text_sum = LatentPSL(WIDTH, 1)
        self.fusion = nn.Parameter(torch.empty(2, WIDTH))
        nn.init.xavier_uniform_(self.fusion, gain=nn.init.calculate_gain('tanh'))

    # ------------------------------------------------------------------ vocabulary projection (Conv1d, k = 1)
    def vocab_matrix(self):
        return self.conv1d.weight.squeeze(-1)                      # (512, V)

    def project(self, x):
        """(B,L,V) logits or any dense caption representation -> (B,L,512)"""
        return _linear(self._cell_ops(x), x, self.vocab_matrix(), self.conv1d.bias)

    def project_ids(self, captions):
        """real captions (B,L) int64: the one-hot product of run_gun.py:447-451 as a gather of weight columns"""
        return F.embedding(captions, self.vocab_matrix().t()) + self.conv1d.bias

    # ------------------------------------------------------------------ everything behind the projection
    def set_ops(self, ops):
        """kernel interface for the fused cell (HipOps; tests: the emulation).  Default: the HIP library for CUDA tensors,
        the plain ATen recurrence for CPU tensors."""
        object.__setattr__(self, '_ops', ops)
        return self

    def _cell_ops(self, x):
        ops = getattr(self, '_ops', None)
        if ops is None and x.is_cuda:
            ops = _hip_ops()                                       # raises if libdlsg_hip.so is missing: no silent fallback on a GPU
        return ops

    def _lstm(self, x, time_major=False):
        """single-layer LSTM, zero initial state, gate order i,f,g,o; unrolled so that autograd can differentiate it twice:
        the recurrent product is an ordinary matmul per step, the cell's pointwise part one fused op per step.
        time_major (kernel path only): return the (L, n, 512) array the recurrence writes instead of its (n, L, 512) view."""
        w_ih, w_hh = self.lstm.weight_ih_l0, self.lstm.weight_hh_l0
        bias = self.lstm.bias_ih_l0 + self.lstm.bias_hh_l0
        n, L, _ = x.shape
        ops = self._cell_ops(x)
        h = x.new_zeros(n, WIDTH)
        c = x.new_zeros(n, WIDTH)
        out = []
        if ops is not None:
            # the whole recurrence as one node per differentiation level (time-major: a step's rows are dense)
            xin = _linear(ops, x.transpose(0, 1).contiguous(), w_ih, bias)
            hs = _LstmSeq.apply(ops, xin, w_hh)[0]
            return hs if time_major else hs.transpose(0, 1)
        xin = F.linear(x, w_ih, bias)                              # all steps' input gates in one product
        for t in range(L):
            i, f, g, o = (xin[:, t] + F.linear(h, w_hh)).chunk(4, dim=1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            h = torch.sigmoid(o) * torch.tanh(c)
            out.append(h)
        return torch.stack(out, 1)

    def _proposal_score(self, m, psl, alpha, words, word_mask):
        """PSLScore2.forward (layer.py:690-715): words (n,L,512) attended by the (top-k) proposals"""
        n = psl.shape[0]
        ops = self._cell_ops(psl)
        lin = lambda layer, x: _linear(ops, x, layer.weight, layer.bias)
        e = _tanh_ln(lin(m.psl_embed[0], psl), m.psl_embed[2], ops)
        if m.select:
            top = alpha.sum(dim=1).topk(m.num_top, dim=-1).indices
            e = e.gather(1, top.unsqueeze(-1).expand(n, m.num_top, WIDTH))
        a = _tanh_ln(lin(m.att_norm[0], words), m.att_norm[2], ops)
        adj = _softmax(_mm(ops, GEMM_NT, a, e, 1.0 / math.sqrt(WIDTH)), 1, ops) * word_mask     # mask after the softmax (:703-704)
        weight = adj.sum(dim=1)
        agg = _dropout(_tanh_ln(_mm(ops, GEMM_TN, adj, a), m.psl_norm[1], ops), 0.3, self.training)
        sc = m.psl_scorer
        pair = lin(sc.classify, torch.tanh(lin(sc.visual_embed[0], e)) * torch.tanh(lin(sc.sent_embed[0], agg))).squeeze(-1)
        return (pair * weight).sum(dim=-1) / weight.sum(dim=-1)                                   # (n,)

    def _proposal_scores(self, psl_o, psl_m, alpha_o, alpha_m, words, word_mask, groups=1):
        """both PSLScore2 heads (object and motion proposals: same shapes, different weights) side by side: every product is one
        batched launch over the two heads, every LayerNorm one grouped launch, every element-wise op one launch on the stacked
        tensor -- half the launches of two `_proposal_score` calls at each differentiation level.  Same arithmetic per head."""
        mo, mm = self.obj_psl_score, self.motion_psl_score
        ops = self._cell_ops(words)
        n, L, _ = words.shape
        P = psl_o.shape[1]
        assert n == groups * psl_o.shape[0]

        def st(f):
            return torch.stack([f(mo), f(mm)])

        def lin(x, layer):
            bias = layer(mo).bias
            return _linear_stacked(ops, x, st(lambda m: layer(m).weight), None if bias is None else st(lambda m: layer(m).bias))

        def ln(x, layer):
            return _tanh_ln_stacked(x, st(lambda m: layer(m).weight), st(lambda m: layer(m).bias), layer(mo).eps, ops)
        # the proposals are the clips' (B rows), the same for every caption set scored against them: embedded once, then repeated
        Bc = psl_o.shape[0]
        e = ln(lin(torch.stack([psl_o, psl_m]).view(2, Bc * P, -1), lambda m: m.psl_embed[0]), lambda m: m.psl_embed[2])
        e = e.view(2, 1, Bc, P, WIDTH).expand(2, groups, Bc, P, WIDTH).reshape(2 * n, P, WIDTH)
        if mo.select:
            top = torch.cat([alpha_o, alpha_m], 0).sum(dim=1).topk(mo.num_top, dim=-1).indices               # (2n, top)
            e = e.gather(1, top.unsqueeze(-1).expand(2 * n, mo.num_top, WIDTH))
        T = e.shape[1]
        a = ln(lin(words.reshape(1, n * L, WIDTH).expand(2, n * L, WIDTH), lambda m: m.att_norm[0]), lambda m: m.att_norm[2])
        a = a.view(2 * n, L, WIDTH)
        adj = _softmax(_mm(ops, GEMM_NT, a, e, 1.0 / math.sqrt(WIDTH)), 1, ops)                            # (2n, L, T)
        adj = (adj.view(2, n, L, T) * word_mask).view(2 * n, L, T)                                          # mask after the softmax
        weight = adj.sum(dim=1)                                                                             # (2n, T)
        agg = _dropout(ln(_mm(ops, GEMM_TN, adj, a).view(2, n * T, WIDTH), lambda m: m.psl_norm[1]), 0.3, self.training)
        v = torch.tanh(lin(e.reshape(2, n * T, WIDTH), lambda m: m.psl_scorer.visual_embed[0]))
        s_ = torch.tanh(lin(agg, lambda m: m.psl_scorer.sent_embed[0]))
        pair = lin(v * s_, lambda m: m.psl_scorer.classify).view(2 * n, T)
        return ((pair * weight).sum(dim=-1) / weight.sum(dim=-1)).view(2, n)                               # [object | motion] x (n,)

    def score_projected(self, h, obj, mot, att_mask, alpha_all, groups=1):
        """h (n,L,512) = projected captions, n = groups * B rows (`groups` caption sets scored against the same clips in one
        pass).  PSLScore2 ends with a mean over ITS batch (layer.py:714: `.mean(axis=-1)` on a (B,) tensor), so the two
        proposal scores are scalars per caption set; the result is (n,)."""
        n = h.shape[0]
        B = n // groups

        def rep(t):
            return t if groups == 1 else t.repeat(groups, *([1] * (t.dim() - 1)))
        x = torch.relu(h)                                          # ResBlock's in-place ReLU also feeds the skip (sublayer.py:111,119)
        conv = self.block[0].res_block[1]
        # Conv1d(512, 512, 3, padding=1) over the word axis as ONE product on the three shifted copies of the sequence
        # (MIOpen's choice for this shape is an im2col + GEMM per sample: 2 x 192 launches per call)
        ops = self._cell_ops(h)
        if ops is not None and hasattr(ops, 'conv_taps') and (x.dtype == torch.float32 or getattr(ops, 'name', '') != 'hip'):
            taps = _Taps.apply(ops, x, False)                                                   # (n, L, 3 x 512): x[t-1] | x[t] | x[t+1]
        else:
            xp = F.pad(x, (0, 0, 1, 1))
            taps = torch.cat([xp[:, :-2], xp[:, 1:-1], xp[:, 2:]], dim=2)
        x = x + 0.3 * _linear(ops, taps, conv.weight.permute(0, 2, 1).reshape(conv.weight.shape[0], -1), conv.bias)
        mask = rep(att_mask)
        sa = self.att
        Lw = x.shape[1]
        if ops is not None:
            # LayerNorm, dropout and the K / Q / V projections are row-wise: they run on the time-major rows the recurrence
            # wrote (no transposed copy of the sequence at any differentiation level); the attention products below read the
            # per-caption (L, 512) blocks as strided views
            y = _dropout(_tanh_ln(self._lstm(x, time_major=True), self.layer_norm, ops, pre_tanh=False), 0.3, self.training)
            rows = y.reshape(1, Lw * n, WIDTH)
        else:
            y = _dropout(_tanh_ln(self._lstm(x), self.layer_norm, ops, pre_tanh=False), 0.3, self.training)
            rows = y.reshape(1, n * Lw, WIDTH)
        # K, Q, V: three same-shape projections of y as one batched product
        kqv = _linear_stacked(ops, rows.expand(3, -1, -1), torch.stack([sa.K.weight, sa.Q.weight, sa.V.weight]))
        kqv = kqv.view(3, Lw, n, -1).transpose(1, 2).unbind(0) if ops is not None else kqv.view(3, n, Lw, -1).unbind(0)
        logits = _mm(ops, GEMM_NT, kqv[0], kqv[1], 1.0 / math.sqrt(sa.attention_size))                    # (n, L, L)
        w = _softmax(torch.where(mask > 0, logits, torch.full_like(logits, -9e15)), -1, ops)
        ctx_ = _linear(ops, _mm(ops, GEMM_NN, w, kqv[2]), sa.output_layer[0].weight)
        words = _tanh_ln(_dropout(ctx_, sa.dropout, self.training), self.att_norm[1], ops)
        word_mask = mask[:, 0, :].unsqueeze(2)                     # (n,L,1)
        alpha = rep(alpha_all) * word_mask
        P = self.num_psl
        # (the two proposal scores and the text summary are independent; recorded on forked streams during the graph capture
        # they replay SLOWER -- 14.8 ms against 11.8 ms per critic update: cross-queue joins cost more than the launch floor saves)
        if ops is not None and self.obj_psl_score.select == self.motion_psl_score.select and obj.shape == mot.shape and \
                not os.environ.get('DLSG_CRITIC_PSL_SEPARATE'):
            both = self._proposal_scores(obj, mot, alpha[:, :, :P], alpha[:, :, -P:], words, word_mask, groups)       # (2, n)
        else:
            both = torch.stack([self._proposal_score(self.obj_psl_score, rep(obj), alpha[:, :, :P], words, word_mask),
                                self._proposal_score(self.motion_psl_score, rep(mot), alpha[:, :, -P:], words, word_mask)])
        both = both.view(2, groups, B).mean(dim=2).repeat_interleave(B, dim=1)                                        # (2, n)
        ts = self.text_sum
        adj = _softmax(_linear(ops, words, ts.theta), 1, ops)      # LatentPSL(512, 1): one latent node over the words
        sent = _dropout(_tanh_ln(_mm(ops, GEMM_TN, adj, words), ts.out_norm[1], ops), 0.3, self.training).squeeze(1)
        fus = _softmax(_linear(ops, sent, self.fusion), -1, ops)
        return (both * fus.t()).sum(dim=0)

    def forward(self, inputs, obj_proposals, motion_proposals, att_mask=None, alpha_all=None):
        return self.score_projected(self.project(inputs), obj_proposals, motion_proposals, att_mask, alpha_all)


def attention_mask(captions):
    """run_gun.py:164-166: (B,L,L) outer product of the non-<pad> mask"""
    seq = (captions > 0).to(torch.float32)
    return seq.unsqueeze(2) * seq.unsqueeze(1)


def critic_step_losses(D, captions, f_caption, obj, mot, att_mask, alpha, eps_gp):
    """run_gun.py:345-375 in one critic pass.  captions (B,L) int64 real ids, f_caption (B,L,V) generator logits (detached),
    eps_gp (B,1,1).  Returns (loss_D, r_loss, f_loss, gradient_penalty, (r_logit, f_logit, mixed_logit))."""
    B = captions.shape[0]
    h_r = D.project_ids(captions)
    h_f = D.project(f_caption)
    h_m = eps_gp * h_r + (1 - eps_gp) * h_f                      # == project(eps * onehot + (1 - eps) * f_caption)
    scores = D.score_projected(torch.cat([h_r, h_f, h_m], 0), obj, mot, att_mask, alpha, groups=3)
    r_logit, f_logit, m_logit = scores[:B], scores[B:2 * B], scores[2 * B:]
    g = torch.autograd.grad(m_logit.sum(), h_m, create_graph=True, retain_graph=True)[0]        # (B,L,512)
    W = D.vocab_matrix()
    ops = D._cell_ops(g)
    gram = _mm(ops, GEMM_NT, W, W)
    # |d mixed_logit / d mixed_captions|_2 per sample.  The floor keeps sqrt's derivative finite where a sample's critic
    # gradient is exactly zero / underflows (torch's .norm(2) of the reference, run_gun.py:366-371, has subgradient 0 there;
    # an unclamped sqrt would put NaN into every critic parameter through the double backward)
    gn = torch.sqrt(((_linear(ops, g, gram)) * g).sum(dim=(1, 2)).clamp_min(1e-24))
    gp = ((gn - 1) * (gn - 1)).mean()
    r_loss, f_loss = r_logit.mean(), f_logit.mean()
    return f_loss - r_loss + 10 * gp, r_loss, f_loss, gp, (r_logit, f_logit, m_logit)


def _adam_shared_count_ok(opt, params):
    """can `_adam_step_shared_count` replace opt.step()?  Plain Adam (what run_gun.py:100 builds) whose state exists on the device
    for exactly `params`, all at the same step count.  Reads the step tensors: call it outside a capture."""
    if len(opt.param_groups) != 1 or not params:
        return False
    g = opt.param_groups[0]
    if g.get('weight_decay', 0) != 0 or g.get('amsgrad', False) or g.get('maximize', False) or g.get('differentiable', False) or \
            torch.is_tensor(g['lr']):
        return False
    with_grad = [p for p in g['params'] if p.grad is not None]
    if len(with_grad) != len(params) or any(a is not b for a, b in zip(with_grad, params)):
        return False
    st = [opt.state.get(p) for p in params]
    if any(s is None or 'exp_avg' not in s or not torch.is_tensor(s['step']) or not s['step'].is_cuda for s in st):
        return False
    steps = torch.stack([s['step'].reshape(()) for s in st])
    return bool((steps == steps[0]).all().item())


def _adam_step_shared_count(opt, params):
    """One Adam step on `opt`'s own state tensors (exp_avg, exp_avg_sq, step: `state_dict()` stays torch.optim.Adam's) with the
    bias corrections computed ONCE from the shared step count.  `torch.optim.Adam(capturable=True)` keeps a 0-dim step tensor per
    parameter and divides every parameter's denominator by ITS OWN bias-correction tensor: lists of big tensors against lists of
    0-dim tensors have no fused foreach path, so the captured step was ~100 broadcast divisions + ~50 scalar launches for the
    critic's 54 parameters (0.45 ms of a 7.8 ms update); this is 12 launches.  Same arithmetic as torch's
    `_multi_tensor_adam` (capturable branch) up to the order of the scalar factors."""
    g = opt.param_groups[0]
    b1, b2 = g['betas']
    lr, eps = g['lr'], g['eps']
    st = [opt.state[p] for p in params]
    grads = [p.grad for p in params]
    avgs, sqs, steps = [s['exp_avg'] for s in st], [s['exp_avg_sq'] for s in st], [s['step'] for s in st]
    with torch.no_grad():
        torch._foreach_add_(steps, 1)
        t = steps[0]                                        # every parameter of the group steps together
        neg_bc1_over_lr = (torch.pow(b1, t) - 1) / lr       # -(1 - b1^t) / lr
        bc2_sqrt = (1 - torch.pow(b2, t)).sqrt()
        torch._foreach_lerp_(avgs, grads, 1 - b1)
        torch._foreach_mul_(sqs, b2)
        torch._foreach_addcmul_(sqs, grads, grads, 1 - b2)
        denom = torch._foreach_sqrt(sqs)
        torch._foreach_div_(denom, bc2_sqrt)
        torch._foreach_add_(denom, eps)
        torch._foreach_mul_(denom, neg_bc1_over_lr)         # p += m / (denom * -(bc1 / lr))  ==  p -= (lr / bc1) m / denom
        torch._foreach_addcdiv_(params, avgs, denom)


class GANLambdaHandler(object):
    """utils/utils.py:196-265: weight of the generator's GAN loss.  Stable at `gan_lambda` while the running caption loss
    (last 200 steps) does not rise; when its newer half exceeds the older half by 4 % the weight follows one half-cosine dip
    down to 0.006 and back over 500 steps.  `cap_list` is what the reference stores in its checkpoints."""
    WIDTH, PERIOD, LOW = 200, 500, 0.006

    def __init__(self, total_step, gan_lambda, cap_list=None):
        self.cap_list = list(cap_list) if cap_list is not None else []
        self.total_step = total_step
        self.current_step = 0
        self.counter = self.PERIOD
        self.current_schedule_step = 0
        self.start_gan_lambda = gan_lambda
        self.low_gan_lambda = self.LOW
        amp = (gan_lambda - self.LOW) / 2
        k = np.arange(self.PERIOD)
        # sin(pi * x / PERIOD) on x in [PERIOD/2, 3 PERIOD/2) resp. [3 PERIOD/2, 5 PERIOD/2): start high -> low -> high
        self.decrease_schedule = (np.sin(np.pi * (k + self.PERIOD // 2) / self.PERIOD) * amp + amp + self.LOW).tolist()
        self.increase_schedule = (np.sin(np.pi * (k + 3 * self.PERIOD // 2) / self.PERIOD) * amp + amp + self.LOW).tolist()
        self.current_lambda = gan_lambda
        self.state = 0                      # 0 stable, 1 decrease, 2 increase (never entered by the reference either)

    def update_gan_lambda(self, epoch, i, cap_loss):
        self.current_step = i - 1 + epoch * self.total_step
        self.cap_list.append(cap_loss)
        if len(self.cap_list) <= self.WIDTH:
            return
        self.cap_list = self.cap_list[-self.WIDTH:]
        if self.state == 0:
            half = self.WIDTH // 2
            if np.mean(self.cap_list[half:]) > 1.04 * np.mean(self.cap_list[:half]):
                self.state = 1
        elif self.current_schedule_step == self.counter - 1:
            self.current_schedule_step = 0
            self.state = 0

    def get_current_lambda(self):
        if self.state != 0:
            sched = self.decrease_schedule if self.state == 1 else self.increase_schedule
            self.current_lambda = sched[self.current_schedule_step]
            self.current_schedule_step += 1
        return self.current_lambda


class GanTrainer(object):
    """One `RunGAN.train` iteration (run_gun.py:147-234) around the HIP generator.

        it = GanTrainer(model, D); out = it.iteration(frames, regions, captions, cap_lens, tf_ratio, epoch, i)

    `trainer` is the generator's `dlsg_amd.Trainer`; on a GPU its step is two hipGraph replays (forward + CrossEntropy |
    backward + Adam) with the critic's term between them, and the critic updates / the critic's term are replayed graphs of
    PyTorch-ROCm launches themselves (use_graphs=False: everything kernel by kernel)."""

    def __init__(self, model, D, lr=1.6e-4, betas=(0.5, 0.9), num_D=5, gan_lambda=0.01, total_step=1, cap_list=None,
                 process_group=None, world_size=1, use_graphs=None):
        from .model import Trainer
        self.model, self.D = model, D
        on_gpu = next(D.parameters()).is_cuda
        # critic updates replayed from a hipGraph (see _critic_graph): needs Adam's step counter on the device
        self.use_graphs = on_gpu if use_graphs is None else (use_graphs and on_gpu)
        # the generator's step replays forward + CrossEntropy and backward + Adam around the GAN term (Trainer.step)
        self.trainer = Trainer(model, lr=lr, betas=betas, process_group=process_group, world_size=world_size,
                               use_graphs=self.use_graphs)
        self.opt_D = torch.optim.Adam(D.parameters(), lr=lr, betas=betas, capturable=self.use_graphs)   # run_gun.py:100
        self.num_D = num_D
        self.lambda_handler = GANLambdaHandler(total_step, gan_lambda, cap_list)
        self.world_size, self.pg = world_size, process_group
        self.eps_source = None              # tests: callable(k) -> (B,1,1) tensor instead of torch.rand
        self._cg, self._cg_seen = {}, set()

    def reset_graphs(self):
        """Drop the captured critic graphs (they hold the Adam state tensors they were captured with: after
        `opt_D.load_state_dict` those are no longer the optimizer's)."""
        self._cg.clear()
        self._cg_seen.clear()

    # ------------------------------------------------------------------ the critic's optimizer (run_gun.py:100)
    def optimizer_d_state_dict(self):
        """`optimizer_d_state_dict` of a checkpoint (run_gun.py:307): torch.optim.Adam layout"""
        sd = self.opt_D.state_dict()
        for st in sd['state'].values():
            st['step'] = st['step'].detach().cpu().reshape(())
        sd['param_groups'][0]['capturable'] = False
        return sd

    def load_optimizer_d_state_dict(self, sd):
        self.opt_D.load_state_dict(sd)
        self.opt_D.param_groups[0]['capturable'] = self.use_graphs
        self.reset_graphs()

    def critic_adam_step(self, grads):
        """one Adam step of the critic's optimizer on given gradients {parameter name: tensor}"""
        for n, p in self.D.named_parameters():
            p.grad = grads[n].to(p.device) if n in grads else None
        self.opt_D.step()

    def _rank_mean(self, value):
        """run_gun.py:433-437 `reduce_tensor`: the mean over ranks of a logged scalar.  Every rank then sees the same caption
        loss, so GANLambdaHandler switches its schedule at the same step everywhere (run_gun.py:202-203,212)."""
        if self.world_size <= 1:
            return float(value)
        import torch.distributed as dist
        t = value.detach().reshape(1).to(torch.float32).clone() if torch.is_tensor(value) else \
            torch.tensor([float(value)], dtype=torch.float32, device=next(self.D.parameters()).device)
        dist.all_reduce(t, group=self.pg)
        return float(t) / self.world_size

    def _allreduce_D(self):
        """mean of the critic's gradients over the ranks (what DDP does for `model_d`, run_gun.py:71-72) as ONE collective on a
        flat copy -- the critic has 54 parameter tensors of 2 KB to 6 MB (15 MB in all): one all-reduce per tensor would be 54
        latency-bound collectives between the two graph replays of every critic update"""
        if self.world_size > 1:
            import torch.distributed as dist
            grads = [p.grad for p in self.D.parameters() if p.grad is not None]
            if not grads:
                return
            flat = torch.cat([g.reshape(-1) for g in grads])
            dist.all_reduce(flat, group=self.pg)
            flat.div_(self.world_size)
            torch._foreach_copy_(grads, [c.view_as(g) for c, g in zip(flat.split([g.numel() for g in grads]), grads)])

    def _critic_graph(self, inputs):
        """One critic update -- the three-way critic pass, the gradient penalty's double backward, the backward of loss_D and
        the Adam step -- is ~3 000 small PyTorch launches (the LSTM is unrolled over 26 steps and differentiated twice):
        26 ms from Python at batch 64.  It is captured once per batch shape into two hipGraphs (losses + backward | Adam,
        the gradient all-reduce of a multi-GPU run sits between them) and replayed; inputs, the penalty's epsilon and the
        parameter gradients live in static buffers.  The first update of a shape runs eagerly (it creates the Adam state and
        the library handles a capture must not create); a changed learning rate or train/eval mode captures again."""
        key = (tuple(tuple(t.shape) for t in inputs), tuple(g['lr'] for g in self.opt_D.param_groups), self.D.training)
        cg = self._cg.get(key)
        if cg is not None:
            return cg
        if key not in self._cg_seen or len(self.opt_D.state) == 0:
            self._cg_seen.add(key)
            return None
        D, opt = self.D, self.opt_D
        dev = inputs[1].device
        st = [t.clone() for t in inputs] + [torch.empty(inputs[0].shape[0], 1, 1, device=dev)]
        params = [p for p in D.parameters() if p.grad is not None]
        out = {}
        torch.cuda.synchronize()
        shared_count = _adam_shared_count_ok(opt, params)
        gA, gB = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(gA, capture_error_mode='thread_local'):   # other threads (RCCL watchdog, loaders) keep working
            # no zero fill + accumulate: with .grad unset the first gradient that reaches a parameter BECOMES its .grad (a
            # buffer of this graph's pool, the same address on every replay) -- one launch less per parameter than adding
            # into a zeroed buffer, ~50 of the ~1 000 launches of an update
            for p in params:
                p.grad = None
            loss_D, r_loss, f_loss, gp, _ = critic_step_losses(D, *st)
            loss_D.backward()
            out['loss_D'], out['w'] = loss_D.detach(), (r_loss - f_loss).detach()
        assert all(p.grad is not None for p in params)
        grads = [p.grad for p in params]
        with torch.cuda.graph(gB, pool=gA.pool(), capture_error_mode='thread_local'):
            if shared_count:
                _adam_step_shared_count(opt, params)
            else:
                opt.step()
        cg = dict(st=st, params=params, grads=grads, out=out, graphs=(gA, gB))
        self._cg[key] = cg
        return cg

    def _generator_term(self, tokens, obj, mot, att_mask, alpha):
        """loss_G = -D(tokens).mean() (run_gun.py:214-217) and d loss_G / d tokens; replayed from a hipGraph like the critic
        updates (first call of a shape: eager)."""
        D = self.D

        def term(tokens, obj, mot, att_mask, alpha):
            tokens = tokens.detach().requires_grad_(True)
            with torch.enable_grad():
                loss_G = -D(tokens, obj, mot, att_mask=att_mask, alpha_all=alpha).mean()
                g = torch.autograd.grad(loss_G, tokens)[0]
            return loss_G.detach(), g
        inputs = (tokens, obj, mot, att_mask, alpha)
        if not self.use_graphs:
            return term(*inputs)
        key = ('G', tuple((tuple(t.shape), tuple(t.stride())) for t in inputs), D.training)
        gg = self._cg.get(key)
        if gg is None:
            if key not in self._cg_seen:
                self._cg_seen.add(key)
                return term(*inputs)
            st = [torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=t.device).copy_(t) for t in inputs]
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                out = term(*st)
            gg = dict(st=st, out=out, graph=graph)
            self._cg[key] = gg
        for dst, src in zip(gg['st'], inputs):
            dst.copy_(src)
        gg['graph'].replay()
        return gg['out']

    def train_disc(self, captions, f_caption, obj, mot, att_mask, alpha):
        """run_gun.py:339-381: num_D critic updates.  Returns (mean loss_D, mean Wasserstein estimate) as floats."""
        mean_loss = torch.zeros((), device=f_caption.device)
        mean_w = torch.zeros((), device=f_caption.device)
        B = captions.shape[0]
        inputs = (captions, f_caption, obj, mot, att_mask, alpha)
        cg = self._critic_graph(inputs) if self.use_graphs else None
        if cg is not None:
            for dst, src in zip(cg['st'], inputs):
                dst.copy_(src)
            for p, g in zip(cg['params'], cg['grads']):
                p.grad = g                    # an eager update in between (another batch shape) re-created the .grad tensors
        for k in range(self.num_D):
            eps = self.eps_source(k) if self.eps_source is not None else torch.rand(B, 1, 1, device=f_caption.device)
            if cg is not None:
                cg['st'][-1].copy_(eps)
                cg['graphs'][0].replay()
                self._allreduce_D()
                cg['graphs'][1].replay()
                mean_loss += cg['out']['loss_D'] / self.num_D
                mean_w += cg['out']['w'] / self.num_D
                continue
            self.opt_D.zero_grad(set_to_none=True)
            loss_D, r_loss, f_loss, gp, _ = critic_step_losses(self.D, captions, f_caption, obj, mot, att_mask, alpha, eps)
            mean_loss += loss_D.detach() / self.num_D
            mean_w += (r_loss.detach() - f_loss.detach()) / self.num_D
            loss_D.backward()
            self._allreduce_D()
            self.opt_D.step()
        return self._rank_mean(mean_loss), self._rank_mean(mean_w)

    def iteration(self, frames, regions, captions, cap_lens, tf_ratio, epoch=0, i=1, max_len=26):
        model, D = self.model, self.D
        captions = captions[:, :max_len].contiguous()
        att_mask = attention_mask(captions)
        # ---- Train D: the generator's outputs are constants here (run_gun.py:167-174)
        fwd = None if os.environ.get('DLSG_GAN_EAGER_FORWARD') else \
            self.trainer.forward_only(frames, regions, captions, tf_ratio, max_len)       # replayed once the step is captured
        if fwd is None:
            with torch.no_grad():
                fwd = model(frames, regions, captions, max_len, tf_ratio)
        f_caption, obj, mot, alpha = fwd
        loss_D, wass = self.train_disc(captions, f_caption, obj, mot, att_mask, alpha)
        # ---- Train the captioning model (run_gun.py:180-234)
        out = {}

        def gan_term(logits_tm, sv):
            """d(gan_lambda * loss_G) / d logits, time-major (L,B,V); called between the HIP forward and backward"""
            s = sv['dec']
            tokens = logits_tm.transpose(0, 1).detach()
            obj_, mot_ = sv['dec_gsrc'][0].detach(), sv['dec_gsrc'][1].detach()
            alpha_ = s['ALPHA'].transpose(0, 1).detach()
            loss_G, g = self._generator_term(tokens, obj_, mot_, att_mask, alpha_)
            out['loss_G'] = loss_G.detach()
            out['cap_loss_dev'] = sv['loss_dev']
            # the reference updates lambda from the caption loss of THIS step before using it (run_gun.py:210,224)
            # -- with several ranks the all-reduced mean, as the reference feeds it (run_gun.py:202-203,212)
            out['cap_loss_record'] = self._rank_mean(sv['loss_dev'])
            self.lambda_handler.update_gan_lambda(epoch, i, out['cap_loss_record'])
            out['gan_lambda'] = self.lambda_handler.get_current_lambda()
            return (g * out['gan_lambda']).transpose(0, 1)
        cap_loss = self.trainer.step(frames, regions, captions, cap_lens, tf_ratio, max_len=max_len, extra_dlogits=gan_term)
        # cap_loss / loss_G: this rank's values (what its backward used); *_record: the means over ranks the reference logs
        out.update(cap_loss=float(cap_loss), loss_G_record=self._rank_mean(out['loss_G']), loss_G=float(out['loss_G']),
                   loss_D=loss_D, wasserstein=wass)
        out['total_loss'] = out['cap_loss'] + out['loss_G'] * out['gan_lambda']
        out.pop('cap_loss_dev')
        check = getattr(model.ops, 'check_persistent', None)
        if check is not None:
            check()                        # the losses above were read back: the device is idle, the time-out words are final
        return out


# ------------------------------------------------------------------ checkpoints (run_gun.py:302-310, 53-61, 92-109)
def save_checkpoint(path, epoch, gan):
    """The dict RunGAN stores: epoch, model_state_dict, optimizer_state_dict (torch.optim.Adam layout), model_d_state_dict,
    optimizer_d_state_dict, cap_list."""
    torch.save({'epoch': epoch,
                'model_state_dict': {k: v.detach().cpu() for k, v in gan.model.state_dict().items()},
                'optimizer_state_dict': gan.trainer.optimizer_state_dict(),
                'model_d_state_dict': {k: v.detach().cpu() for k, v in gan.D.state_dict().items()},
                'optimizer_d_state_dict': gan.optimizer_d_state_dict(),
                'cap_list': np.array(gan.lambda_handler.cap_list)}, path)


def load_checkpoint(path, gan, map_location=None):
    """Resume a GanTrainer from a checkpoint written by save_checkpoint or by the reference trainer.  Returns the epoch."""
    ck = torch.load(path, map_location=map_location or 'cpu', weights_only=False)
    gan.model.load_state_dict(ck['model_state_dict'])
    gan.trainer.load_optimizer_state_dict(ck['optimizer_state_dict'])
    gan.D.load_state_dict(ck['model_d_state_dict'])
    gan.load_optimizer_d_state_dict(ck['optimizer_d_state_dict'])
    h = gan.lambda_handler
    gan.lambda_handler = GANLambdaHandler(h.total_step, h.start_gan_lambda, cap_list=ck['cap_list'])
    return ck['epoch']
