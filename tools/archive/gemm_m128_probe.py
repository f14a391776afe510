"""Recurrent gate products at batch 128 per GPU (M = 128): 64x64 tile vs 128x128 tile, contraction split over slabs.
usage: python3 tools/archive/gemm_m128_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
import torch  # noqa: E402
from dlsg_amd.hip import HipOps, GEMM_NT, GEMM_NN  # noqa: E402

ops = HipOps()
for x3, nm in ((0, 'fp32'), (1024, 'bf16x3')):
    for mode, mn, M, N, K in ((GEMM_NT, 'NT', 128, 4096, 4096), (GEMM_NT, 'NT', 128, 4096, 2348), (GEMM_NN, 'NN', 128, 4096, 4096),
                              (GEMM_NN, 'NN', 128, 2048, 4096)):
        g = torch.Generator().manual_seed(0)
        A = torch.randn(M, K, generator=g).cuda()
        B = (torch.randn(N, K, generator=g) if mode == GEMM_NT else torch.randn(K, N, generator=g)).cuda()
        line = '%s %s %d %d %d |' % (nm, mn, M, N, K)
        for force, ksl in ((256, (1, 2, 3, 4)), (512, (2, 4, 8, 12, 16))):
            for ks in ksl:
                step = (K // ks + 31) // 32 * 32
                kb = [(k, min(K, k + step)) for k in range(0, K, step)]
                slabs = torch.empty(len(kb), M, N, device='cuda')

                def go():
                    ops.gemm(mode, [(A[:, k0:k1], B[:, k0:k1] if mode == GEMM_NT else B[k0:k1], slabs[i]) for i, (k0, k1) in enumerate(kb)],
                             flags=force | x3)
                for _ in range(2):
                    go()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    go()
                e1.record()
                torch.cuda.synchronize()
                line += ' %s/ks%d %.0fus' % ('64' if force == 256 else '128', len(kb), e0.elapsed_time(e1) / 10 * 1e3)
        print(line)
