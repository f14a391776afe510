"""The graph-attention pass of SURVEY.md 8(d) in isolation: everything between the dense projections of one forward --
object->frame graph (x2 streams), LatentPSL (x2), the 26x26 self-attention core, and the decoder attention over the
cached K', V' for every word step -- on synthetic operands of the MSVD shape, timed with HIP events, with the kernels
the train step and the beam search launch (`o2v_fwd`, `latent_psl_fwd`, `sa_core_fwd`, and `dec_mid_fwd`, the fused word
step whose attention phase reads the K', V' cache).
Algorithmic bytes per clip follow SURVEY.md 8(d): 8.79 MB (MSVD-shaped) = 4.96 MB that stream from HBM once (encoder
graphs) + 3.83 MB of K', V' reads that repeat 26x on the same 134 MB (1024 clips) and are served by the Infinity Cache;
the two parts are reported separately and only the first is held against the HBM roofline."""
import math

import torch

from .hip import GEMM_NT, GEMM_NN, GEMM_TN


def pass_bytes_per_clip(T, O, H, P, L, ns=2, parts=False):
    o2v = ns * 4 * (T * O * H + 2 * T * H)
    psl = ns * 4 * (T * H + P * H)
    sa = 4 * 4 * (T * 2 * H)
    dec = ns * L * 4 * (2 * P * H + 2 * H)
    if parts:
        return {'o2v': o2v, 'latent_psl': psl, 'self_attention_core': sa, 'decoder_attention': dec}
    return o2v + psl + sa + dec


def run_graph_attention_pass(ops, B=1024, T=26, O=16, H=1024, P=8, L=26, R=2048, reps=3, device='cuda'):
    g = torch.Generator(device=device).manual_seed(0)

    def r(*shape, scale=1.0):
        return torch.randn(*shape, device=device, generator=g) * scale

    NO = T * O
    ys = [torch.tanh(r(B, NO, H)) for _ in range(2)]
    vs = [r(B, T, H) for _ in range(2)]
    gam, bet = torch.ones(H, device=device), torch.zeros(H, device=device)
    theta = r(P, H, scale=0.05)
    z = [torch.empty(B * T, H, device=device) for _ in range(2)]
    ml = torch.empty(B * T, 2, device=device)
    ost = torch.empty(B * NO, 2, device=device)
    S = torch.empty(B, NO, T, device=device)
    lg = torch.empty(B, T, P, device=device); adj = torch.empty(B, T, P, device=device)
    u = torch.empty(B * P, H, device=device); psl = torch.empty(B * P, H, device=device); stp = torch.empty(B * P, 2, device=device)
    D2 = 2 * H
    Kp, Qp, Vp = r(B, T, D2, scale=0.1), r(B, T, D2, scale=0.1), r(B, T, D2)
    slg = torch.empty(B, T, T, device=device); sw = torch.empty(B, T, T, device=device); att = torch.empty(B, T, D2, device=device)
    Kc = [r(B, P, H, scale=0.1) for _ in range(2)]; Vc = [r(B, P, H) for _ in range(2)]
    # operands of the fused word step (csrc/decstep.hip dec_mid_fwd): query-cell gate slabs, cell state, LayerNorms
    Q = H
    slabs = r(1, B, 4 * Q, scale=0.5)
    gq = r(B, 4 * Q, scale=0.1)
    b4 = torch.zeros(4 * Q, device=device)
    c_prev = r(B, Q, scale=0.1)
    c_new, h_new, gates = torch.empty(B, Q, device=device), torch.empty(B, Q, device=device), torch.empty(B, 4 * Q, device=device)
    qcur, st_q = torch.empty(B, Q, device=device), torch.empty(B, 2, device=device)
    cpre = [torch.empty(B, H, device=device) for _ in range(2)]
    ctx = [torch.empty(B, H, device=device) for _ in range(2)]
    st_c = [torch.empty(B, 2, device=device) for _ in range(2)]
    alpha = torch.empty(B, 2 * P, device=device)
    tiles = (NO + 15) // 16
    nsplit = max(1, min(tiles, 256 // (2 * B)))
    sc = 1.0 / math.sqrt(R)

    marks = []

    def mark(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append((name, e))

    def one_pass():
        mark('start')
        # both encoder streams in one launch, as CapGnnModel._encode issues it
        ops.o2v_fwd_multi([dict(y=ys[i], v=vs[i], g_obj=gam, b_obj=bet, z=z[i], ml=ml, ostats=ost, S=S) for i in range(2)], sc, nsplit)
        mark('o2v')
        for i in range(2):
            ops.latent_psl_fwd(z[i].view(B, T, H), theta, gam, bet, adj, u, psl, stp)
        mark('latent_psl')
        ops.sa_core_fwd(Kp, Qp, Vp, sw, att, 1.0 / math.sqrt(D2))
        mark('self_attention_core')
        for _ in range(L):
            ops.dec_mid_fwd(slabs, gq, b4, b4, c_prev, c_new, h_new, gates, (gam, bet), qcur, st_q, 0.0, 0, Kc, Vc,
                            [(gam, bet), (gam, bet)], cpre, ctx, st_c, alpha, [0.0, 0.0], [0, 0], 1.0 / math.sqrt(H))
        mark('decoder_attention')

    one_pass()
    torch.cuda.synchronize()
    del marks[:]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        one_pass()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nbytes = pass_bytes_per_clip(T, O, H, P, L) * B
    parts = {}
    for (n0, ev0), (n1, ev1) in zip(marks[:-1], marks[1:]):
        if n1 != 'start':
            parts[n1] = parts.get(n1, 0.0) + ev0.elapsed_time(ev1) / reps
    pb = pass_bytes_per_clip(T, O, H, P, L, parts=True)
    enc_ms = parts['o2v'] + parts['latent_psl'] + parts['self_attention_core']
    enc_bytes = (pb['o2v'] + pb['latent_psl'] + pb['self_attention_core']) * B
    return {'clips': B, 'ms': round(ms, 3), 'parts_ms': {k: round(v, 3) for k, v in parts.items()}, 'bytes_per_clip': pass_bytes_per_clip(T, O, H, P, L),
            'achieved_GBps': round(nbytes / ms / 1e6, 1), 'clips_per_s': round(B / ms * 1e3, 0),
            'parts_GBps': {k: round(pb[k] * B / v / 1e6, 1) for k, v in parts.items() if v > 0},
            'hbm_streaming_part': {'bytes_per_clip': enc_bytes // B, 'ms': round(enc_ms, 3),
                                   'achieved_GBps': round(enc_bytes / enc_ms / 1e6, 1),
                                   'clips_per_s': round(B / enc_ms * 1e3, 0)},
            'decoder_term': {'kernel': 'dec_mid_fwd x %d (fused word step: query-cell pointwise + LayerNorm + attention over '
                                       "the K', V' cache of both streams + tanh + LayerNorm)" % L,
                             'bytes_per_clip': pb['decoder_attention'], 'ms': round(parts['decoder_attention'], 3),
                             'achieved_GBps': round(pb['decoder_attention'] * B / parts['decoder_attention'] / 1e6, 1),
                             'note': "K', V' of %d clips = %.0f MB, re-read every word step: Infinity-Cache resident, not an HBM "
                                     'stream' % (B, 2 * 2 * B * P * H * 4 / 1e6)}}
