"""The train step's mid-size products (1 664 = 26 frames x 64 clips rows, 512-6 144 wide: 3.4 ms of the step at 55-114 TFLOP/s,
tools/gemm_census.py) under every tile the library has and under a K split into groups writing slabs + one slab_reduce:
which launches would gain from a different dispatch.  usage: python tools/archive/gemm_mid_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
import torch  # noqa: E402
from dlsg_amd.hip import HipOps, GEMM_NT, GEMM_NN, GEMM_TN, F_FORCE64, F_FORCE128  # noqa: E402

ops = HipOps()
NAMES = {GEMM_NT: 'nt', GEMM_NN: 'nn', GEMM_TN: 'tn'}
TILES = {'auto': 0, '64x64': F_FORCE64, '128x64': F_FORCE64 | F_FORCE128, '128x128': F_FORCE128}


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3


SHAPES = [(GEMM_NN, 1664, 1024, 4096, 1), (GEMM_NT, 1664, 4096, 1024, 2), (GEMM_NT, 1664, 1024, 6144, 1), (GEMM_TN, 1024, 6144, 1664, 1),
          (GEMM_NT, 1664, 1024, 2048, 1), (GEMM_TN, 1024, 2048, 1664, 1), (GEMM_TN, 1024, 1024, 512, 8), (GEMM_NN, 512, 1024, 1024, 2),
          (GEMM_NT, 1664, 2048, 2048, 3), (GEMM_TN, 2048, 2048, 1664, 3), (GEMM_NN, 1664, 2048, 2048, 3), (GEMM_NN, 1664, 2048, 1024, 1),
          (GEMM_TN, 4096, 2048, 64, 1), (GEMM_TN, 1000, 1024, 1664, 1), (GEMM_NT, 1664, 1000, 1024, 1), (GEMM_NN, 1664, 1024, 1000, 1)]
for mode, M, N, K, G in SHAPES:
    A = [torch.randn(*((K, M) if mode == GEMM_TN else (M, K)), device='cuda') for _ in range(G)]
    B = [torch.randn(*((N, K) if mode == GEMM_NT else (K, N)), device='cuda') for _ in range(G)]
    C = [torch.empty(M, N, device='cuda') for _ in range(G)]
    flop = 2.0 * M * N * K * G
    res = {}
    for name, fl in TILES.items():
        res[name] = timed(lambda: ops.gemm(mode, list(zip(A, B, C)), flags=fl))
    for ks in (2, 3, 4):
        if G * ks > 16 or K // ks < 256:
            continue
        step = ((K + ks - 1) // ks + 31) // 32 * 32
        bounds = [(k, min(K, k + step)) for k in range(0, K, step)]
        slabs = [torch.empty(len(bounds), M, N, device='cuda') for _ in range(G)]

        def cut(t, is_a, k0, k1):
            if mode == GEMM_TN:
                return t[k0:k1]
            if is_a:
                return t[:, k0:k1]
            return t[:, k0:k1] if mode == GEMM_NT else t[k0:k1]
        groups = [(cut(A[g], True, k0, k1), cut(B[g], False, k0, k1), slabs[g][i]) for g in range(G) for i, (k0, k1) in enumerate(bounds)]
        for name, fl in (('auto', 0), ('128x64', F_FORCE64 | F_FORCE128), ('128x128', F_FORCE128)):
            def run():
                ops.gemm(mode, groups, flags=fl)
                for g in range(G):
                    ops.slab_reduce(slabs[g], C[g])
            res['k%d/%s' % (ks, name)] = timed(run)
    best = min(res, key=res.get)
    print('%s %5d x %5d x %5d g%d: ' % (NAMES[mode], M, N, K, G) + '  '.join('%s %.0f' % (k, v) for k, v in res.items()) +
          '   | best %s %.0f us = %.0f TFLOP/s (auto %.0f)' % (best, res[best], flop / res[best] / 1e6, flop / res['auto'] / 1e6))
