"""GPU microbenchmark of dlsg_gemm over the GEMM shapes of the batch-64 MSVD-shaped train step, per block-tile config.
Usage (GPU box): python tools/archive/gemm_bench.py > gpurun_out/gemm_bench.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
import torch  # noqa: E402
from dlsg_amd.hip import HipOps, GEMM_NT, GEMM_NN, GEMM_TN  # noqa: E402

ops = HipOps()
dev = 'cuda'
SHAPES = [
    ('NT', 4096, 4096, 4096, 1), ('NN', 4096, 4096, 4096, 1), ('TN', 4096, 4096, 4096, 1),
    ('NT', 26624, 1024, 2048, 1), ('NT', 1664, 1024, 2048, 1), ('NT', 1664, 1024, 6144, 1), ('NT', 1664, 4096, 1024, 2),
    ('NT', 1664, 2048, 2048, 3), ('NT', 1664, 1000, 1024, 1), ('NT', 512, 1024, 1024, 2), ('NT', 64, 4096, 2348, 1),
    ('NT', 64, 4096, 4096, 1),
    ('NN', 1664, 1024, 1000, 1), ('NN', 1664, 2048, 2048, 1), ('NN', 1664, 1024, 4096, 1), ('NN', 64, 3072, 4096, 1),
    ('NN', 64, 1024, 4096, 1),
    ('TN', 1024, 2048, 26624, 1), ('TN', 1024, 2048, 1664, 1), ('TN', 1024, 6144, 1664, 1), ('TN', 4096, 1024, 1664, 1),
    ('TN', 2048, 2048, 1664, 1), ('TN', 4096, 300, 1664, 1), ('TN', 1000, 1024, 1664, 1), ('TN', 4096, 2048, 64, 1),
]
MODE = {'NT': GEMM_NT, 'NN': GEMM_NN, 'TN': GEMM_TN}


def run(mode, M, N, K, G, force, ksplit=1):
    g = torch.Generator(device='cpu').manual_seed(0)
    if mode == 'NT':
        A, B = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    elif mode == 'NN':
        A, B = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g)
    else:
        A, B = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g)
    A, B = A.to(dev), B.to(dev)
    groups = []
    nsl = G * ksplit
    C = torch.empty(nsl, M, N, device=dev)
    kb = [(i * K // ksplit, (i + 1) * K // ksplit) for i in range(ksplit)]
    for gi in range(G):
        for i, (k0, k1) in enumerate(kb):
            if mode == 'NT':
                groups.append((A[:, k0:k1], B[:, k0:k1], C[gi * ksplit + i]))
            elif mode == 'NN':
                groups.append((A[:, k0:k1], B[k0:k1], C[gi * ksplit + i]))
            else:
                groups.append((A[k0:k1], B[k0:k1], C[gi * ksplit + i]))
    for _ in range(2):
        ops.gemm(MODE[mode], groups, flags=force)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 5
    e0.record()
    for _ in range(n):
        ops.gemm(MODE[mode], groups, flags=force)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    return ms, 2.0 * M * N * K * G / ms / 1e9


print('%-4s %6s %6s %6s %2s | %22s | %22s | %s' % ('mode', 'M', 'N', 'K', 'G', '64x64 ms / TF', '128x128 ms / TF', 'ksplit variants (128 tile unless noted)'))
for mode, M, N, K, G in SHAPES:
    r64 = run(mode, M, N, K, G, 256)
    r128 = run(mode, M, N, K, G, 512)
    extra = ''
    if M > 64:
        x64 = run(mode, M, N, K, G, 1024 | 256)
        x128 = run(mode, M, N, K, G, 1024 | 512)
        extra += ' X3 64: %.3f/%.0fTF 128: %.3f/%.0fTF' % (x64[0], x64[1], x128[0], x128[1])
    if M <= 64:
        for ks in (1, 2, 4, 8, 16):
            if G * ks <= 16:
                a = run(mode, M, N, K, G, 0, ks)
                b = run(mode, M, N, K, G, 1024, ks)
                extra += ' ks%d: f32 %.3f/%.0fTF x3 %.3f/%.0fTF' % (ks, a[0], a[1], b[0], b[1])
    if False and K >= 1664 and M * N <= 4096 * 2048 * 2:
        for ks in (2, 4, 8):
            if G * ks <= 16:
                a = run(mode, M, N, K, G, 512, ks)
                b = run(mode, M, N, K, G, 256, ks)
                extra += ' ks%d: %.3f/%.0fTF (64: %.3f/%.0fTF)' % (ks, a[0], a[1], b[0], b[1])
    print('%-4s %6d %6d %6d %2d | %9.3f ms %7.1f TF | %9.3f ms %7.1f TF |%s' % (mode, M, N, K, G, r64[0], r64[1], r128[0], r128[1], extra))
    sys.stdout.flush()
