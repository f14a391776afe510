import os, sys
sys.path.insert(0, 'd-lsg-video-caption_amd')
import torch
from dlsg_amd.hip import HipOps, GEMM_NT, GEMM_NN
ops = HipOps()
def run(mode, M, N, K, ks, flags):
    g = torch.Generator().manual_seed(0)
    A = torch.randn(M, K, generator=g).cuda()
    B = (torch.randn(N, K, generator=g) if mode == GEMM_NT else torch.randn(K, N, generator=g)).cuda()
    C = torch.empty(ks, M, N, device='cuda')
    step = (K // ks + 127) // 128 * 128
    kb = [(k, min(K, k + step)) for k in range(0, K, step)]
    groups = [(A[:, k0:k1], B[:, k0:k1] if mode == GEMM_NT else B[k0:k1], C[i]) for i, (k0, k1) in enumerate(kb)]
    for _ in range(3): ops.gemm(mode, groups, flags=flags)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.gemm(mode, groups, flags=flags)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3
for M, N, K in ((64, 4096, 4096), (64, 2048, 4096), (64, 2048, 4096)):
    for ks in (3, 4, 6, 8):
        print('M%d N%d K%d ks%d | fp32 NT %.1f NN %.1f | x3 NT %.1f NN %.1f us' % (M, N, K, ks, run(GEMM_NT, M, N, K, ks, 0), run(GEMM_NN, M, N, K, ks, 0), run(GEMM_NT, M, N, K, ks, 1024), run(GEMM_NN, M, N, K, ks, 1024)))
