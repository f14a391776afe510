"""obj_embed weight gradient (1024 x 2048, contraction 26624): row-group split of the contraction, 128x128 split-bf16 tiles.
usage: python3 tools/archive/gemm_deep_tn_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
import torch  # noqa: E402
from dlsg_amd.hip import HipOps, GEMM_TN  # noqa: E402

ops = HipOps()
M, N, K = 1024, 2048, 26624
g = torch.Generator().manual_seed(0)
A = torch.randn(K, M, generator=g).cuda()
B = torch.randn(K, N, generator=g).cuda()
out = torch.zeros(M, N, device='cuda')
for x3, nm in ((1024, 'bf16x3'), (0, 'fp32')):
    line = nm + ':'
    for force in (512, 256):
        for ks in (1, 4, 6, 8, 12, 16):
            step = (K // ks + 31) // 32 * 32
            kb = [(k, min(K, k + step)) for k in range(0, K, step)]
            slabs = torch.empty(len(kb), M, N, device='cuda')

            def go():
                if len(kb) == 1:
                    ops.gemm(GEMM_TN, [(A, B, out)], flags=force | x3 | 1)
                else:
                    ops.gemm(GEMM_TN, [(A[k0:k1], B[k0:k1], slabs[i]) for i, (k0, k1) in enumerate(kb)], flags=force | x3)
                    ops.slab_reduce(slabs, out, flags=1)
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
            line += ' %s/ks%d %.0fus' % ('128' if force == 512 else '64', len(kb), ms * 1e3)
    print(line)
