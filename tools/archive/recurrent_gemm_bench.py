"""The recurrent (M = batch = 64) gate products of one word step / BiLSTM step as the engine launches them, timed in
isolation: grouped NT launches writing K-split slabs (forward) and grouped NN launches (input gradients).
    python tools/archive/recurrent_gemm_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
import torch  # noqa: E402
from dlsg_amd.hip import HipOps, GEMM_NT, GEMM_NN  # noqa: E402

ops = HipOps()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.Generator(device='cuda').manual_seed(0)


def r(*s):
    return torch.randn(*s, device='cuda', generator=g)


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def nt_case(name, N, segs):
    W = r(N, sum(segs))
    xs = [r(B, k) for k in segs]
    chunk = int(os.environ.get('KCHUNK', '1024'))
    pieces, c0 = [], 0
    for x, k in zip(xs, segs):
        n = max(1, (k + chunk - 1) // chunk)
        step = ((k + n - 1) // n + 15) // 16 * 16
        for k0 in range(0, k, step):
            k1 = min(k, k0 + step)
            pieces.append((x[:, k0:k1], W[:, c0 + k0:c0 + k1]))
        c0 += k
    if os.environ.get('BIGFIRST'):
        pieces.sort(key=lambda ab: -ab[0].shape[1])
    slabs = torch.empty(len(pieces), B, N, device='cuda')
    groups = [(a_, b_, slabs[i]) for i, (a_, b_) in enumerate(pieces)]
    us = timeit(lambda: ops.gemm(GEMM_NT, groups))
    gf = 2.0 * B * N * sum(segs) / 1e9
    print('%-34s NT  %6.1f us  %6.1f TFLOP/s  (%.2f GFLOP, weights %.0f MB)' % (name, us, gf / us, gf, N * sum(segs) * 4 / 1e6))
    ref = sum(x @ W[:, c:c + k].t() for x, k, c in zip(xs, segs, [sum(segs[:i]) for i in range(len(segs))]))
    err = (slabs.sum(0) - ref).abs().max().item() / ref.abs().max().item()
    assert err < 1e-5, err


def nn_case(name, Kc, widths, chunk=1024):
    dy = r(B, Kc)
    Ws = [r(Kc, wd) for wd in widths]
    tot = sum(widths)
    bounds = [(k, min(Kc, k + chunk)) for k in range(0, Kc, chunk)]
    slabs = torch.empty(len(bounds), B, tot, device='cuda')
    groups, c0 = [], 0
    for Wm, wd in zip(Ws, widths):
        for i, (k0, k1) in enumerate(bounds):
            groups.append((dy[:, k0:k1], Wm[k0:k1, :], slabs[i][:, c0:c0 + wd]))
        c0 += wd
    us = timeit(lambda: ops.gemm(GEMM_NN, groups))
    gf = 2.0 * B * Kc * tot / 1e9
    print('%-34s NN  %6.1f us  %6.1f TFLOP/s  (%.2f GFLOP)' % (name, us, gf / us, gf))


nt_case('query gates  [word|lang_h|q_h]', 4096, [300, 1024, 1024])
nt_case('lang gates   [ctx|ctx2|q|lang_h]', 4096, [1024, 1024, 1024, 1024])
nt_case('BiLSTM step, both directions', 8192, [1024])
nn_case('d lang-cell inputs [W_ih|W_hh]', 4096, [3072, 1024])
nn_case('d query-cell rec   [W_hh|W_lang]', 4096, [1024, 1024])
nn_case('BiLSTM rec grad, two dirs', 4096, [1024, 1024])
