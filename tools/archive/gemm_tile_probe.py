"""Block-tile probe for the mid-size products of the train step (M = 1664 = 26 frames x 64 clips and the weight gradients
over them): 64x64, 128x64 (FORCE64|FORCE128; BK from DLSG_GEMM_12864_BK) and 128x128 tiles, ms and TFLOP/s per launch.
usage: python tools/archive/gemm_tile_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
import torch  # noqa: E402
from dlsg_amd.hip import HipOps, GEMM_NT, GEMM_NN, GEMM_TN  # noqa: E402

ops = HipOps()
MODE = {'NT': GEMM_NT, 'NN': GEMM_NN, 'TN': GEMM_TN}
SHAPES = [('NT', 1664, 1024, 2048, 1), ('NT', 1664, 1024, 6144, 1), ('NT', 1664, 4096, 1024, 2), ('NT', 1664, 2048, 2048, 3),
          ('NT', 1664, 1000, 1024, 1), ('NT', 512, 1024, 1024, 4),
          ('TN', 4096, 1024, 1664, 7), ('TN', 4096, 1024, 1664, 4), ('TN', 2048, 2048, 1664, 3), ('TN', 1024, 6144, 1664, 1),
          ('TN', 1024, 2048, 1664, 1), ('TN', 1000, 1024, 1664, 1), ('TN', 1024, 1024, 512, 4),
          ('NN', 1664, 2048, 2048, 3), ('NN', 1664, 1024, 4096, 1), ('NN', 1664, 2048, 1024, 1), ('NN', 1664, 1024, 1000, 1),
          ('NN', 512, 1024, 1024, 2)]


def run(mode, M, N, K, G, force):
    g = torch.Generator().manual_seed(0)
    groups = []
    for _ in range(G):
        if mode == 'NT':
            A, B = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
        elif mode == 'NN':
            A, B = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g)
        else:
            A, B = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g)
        groups.append((A.cuda(), B.cuda(), torch.empty(M, N, device='cuda')))
    for _ in range(3):
        ops.gemm(MODE[mode], groups, flags=force)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        ops.gemm(MODE[mode], groups, flags=force)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    return ms * 1e3, 2.0 * M * N * K * G / ms / 1e9, groups[0][2]


print('BK of the 128x64 tile:', os.environ.get('DLSG_GEMM_12864_BK', '32'))
print('%-3s %5s %5s %5s %2s | %16s | %16s | %16s | default' % ('op', 'M', 'N', 'K', 'G', '64x64 us / TF', '128x64 us / TF', '128x128 us / TF'))
for mode, M, N, K, G in SHAPES:
    r = [run(mode, M, N, K, G, f) for f in (256, 768, 512, 0)]
    assert torch.allclose(r[0][2], r[1][2], rtol=1e-4, atol=1e-3)
    print('%-3s %5d %5d %5d %2d | %8.1f %7.1f | %8.1f %7.1f | %8.1f %7.1f | %8.1f %7.1f' %
          ((mode, M, N, K, G) + sum(((x[0], x[1]) for x in r), ())))
    sys.stdout.flush()
