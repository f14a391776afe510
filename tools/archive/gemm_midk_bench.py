"""Mid-size (1664-row) forward GEMMs of the batch-64 step: 64x64 tile vs 128x128 tile with the contraction split over
slabs (+ the slab_reduce that folds them).  usage: python3 tools/archive/gemm_midk_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
import torch  # noqa: E402
from dlsg_amd.hip import HipOps, GEMM_NT, GEMM_NN, GEMM_TN  # noqa: E402

ops = HipOps()
dev = 'cuda'
SHAPES = [('NT', 1664, 1024, 6144, 1), ('NT', 1664, 1024, 2048, 2), ('NT', 1664, 4096, 1024, 2), ('NT', 1664, 2048, 2048, 3),
          ('NT', 1664, 1000, 1024, 1), ('NT', 1664, 1024, 2048, 1), ('NN', 1664, 1024, 4096, 1), ('NN', 1664, 2048, 2048, 1),
          ('TN', 4096, 1024, 1664, 1), ('TN', 2048, 2048, 1664, 1), ('TN', 1024, 6144, 1664, 1), ('TN', 4096, 300, 1664, 1)]
MODE = {'NT': GEMM_NT, 'NN': GEMM_NN, 'TN': GEMM_TN}


def run(mode, M, N, K, G, force, ks, x3=0):
    g = torch.Generator().manual_seed(0)
    A = (torch.randn(M, K, generator=g) if mode != 'TN' else torch.randn(K, M, generator=g)).to(dev)
    Bs = [(torch.randn(N, K, generator=g) if mode == 'NT' else torch.randn(K, N, generator=g)).to(dev) for _ in range(G)]
    out = torch.empty(G, M, N, device=dev)
    slabs = torch.empty(G, ks, M, N, device=dev)
    step = (K // ks + 31) // 32 * 32
    kb = [(k, min(K, k + step)) for k in range(0, K, step)]

    def go():
        groups = []
        for gi in range(G):
            for i, (k0, k1) in enumerate(kb):
                dst = out[gi] if len(kb) == 1 else slabs[gi][i]
                if mode == 'TN':
                    groups.append((A[k0:k1], Bs[gi][k0:k1], dst))
                else:
                    groups.append((A[:, k0:k1], Bs[gi][:, k0:k1] if mode == 'NT' else Bs[gi][k0:k1], dst))
        ops.gemm(MODE[mode], groups, flags=force | x3)
        if len(kb) > 1:
            for gi in range(G):
                ops.slab_reduce(slabs[gi][:len(kb)], out[gi])
    for _ in range(2):
        go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        go()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    return ms, 2.0 * M * N * K * G / ms / 1e9


for x3, nm in ((1024, 'bf16x3'),):
    for mode, M, N, K, G in SHAPES:
        line = '%s %-2s %5d %5d %5d G%d |' % (nm, mode, M, N, K, G)
        for force, ks in ((256, 1), (512, 1), (512, 2), (512, 3), (512, 4), (512, 5), (512, 8), (256, 2)):
            if G * ks > 16:
                continue
            ms, tf = run(mode, M, N, K, G, force, ks, x3)
            line += ' %s/ks%d %.0fus %.0fTF |' % ('64' if force == 256 else '128', ks, ms * 1e3, tf)
        print(line)
        sys.stdout.flush()
