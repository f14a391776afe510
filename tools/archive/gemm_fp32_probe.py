import os, sys
sys.path.insert(0, 'd-lsg-video-caption_amd')
import torch
from dlsg_amd.hip import HipOps, GEMM_NT, GEMM_NN, GEMM_TN
ops = HipOps()
def run(mode, M, N, K, G, flags):
    g = torch.Generator().manual_seed(0)
    A = (torch.randn(M, K, generator=g) if mode != GEMM_TN else torch.randn(K, M, generator=g)).cuda()
    Bs = [(torch.randn(N, K, generator=g) if mode == GEMM_NT else torch.randn(K, N, generator=g)).cuda() for _ in range(G)]
    C = torch.empty(G, M, N, device='cuda')
    groups = [(A, Bs[i], C[i]) for i in range(G)]
    for _ in range(2): ops.gemm(mode, groups, flags=flags)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.gemm(mode, groups, flags=flags)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    return ms, 2.0 * M * N * K * G / ms / 1e9
for name, mode, M, N, K, G in [('NT', GEMM_NT, 26624, 1024, 2048, 2), ('NT', GEMM_NT, 4096, 4096, 4096, 1), ('NT', GEMM_NT, 1664, 2048, 2048, 3),
                               ('NT', GEMM_NT, 1664, 1024, 6144, 1), ('NT', GEMM_NT, 1664, 4096, 1024, 2), ('NN', GEMM_NN, 4096, 4096, 4096, 1), ('TN', GEMM_TN, 4096, 4096, 4096, 1), ('TN', GEMM_TN, 1024, 2048, 26624, 1), ('NN', GEMM_NN, 26624, 2048, 1024, 1)]:
    r128 = run(mode, M, N, K, G, 512); r64 = run(mode, M, N, K, G, 256)
    x128 = run(mode, M, N, K, G, 1024 | 512); x64 = run(mode, M, N, K, G, 1024 | 256)
    print('%s %6d %5d %5d G%d | 128: %.3f ms %.1f TF | 64: %.3f ms %.1f TF | x3 128: %.3f ms %.0f TF | x3 64: %.3f ms %.0f TF' % (name, M, N, K, G, r128[0], r128[1], r64[0], r64[1], x128[0], x128[1], x64[0], x64[1]))
