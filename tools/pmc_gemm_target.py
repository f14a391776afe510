"""Target for a rocprofv3 --pmc pass over the GEMM variants whose behaviour needs explaining."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
import torch  # noqa: E402
from dlsg_amd.hip import HipOps, GEMM_NT, GEMM_NN, F_BF16X3  # noqa: E402

ops = HipOps()
dev = 'cuda'


def run(mode, M, N, K, ks, flags):
    A = torch.randn(M, K, device=dev)
    B = torch.randn(N, K, device=dev) if mode == GEMM_NT else torch.randn(K, N, device=dev)
    C = torch.empty(ks, M, N, device=dev)
    step = K // ks
    groups = []
    for i in range(ks):
        k0, k1 = i * step, (i + 1) * step
        groups.append((A[:, k0:k1], B[:, k0:k1] if mode == GEMM_NT else B[k0:k1], C[i]))
    for _ in range(3):
        ops.gemm(mode, groups, flags=flags)


run(GEMM_NT, 64, 4096, 4096, 4, 0)            # skinny fp32
run(GEMM_NT, 64, 4096, 4096, 4, F_BF16X3)     # skinny x3
run(GEMM_NN, 64, 4096, 4096, 4, F_BF16X3)     # skinny x3 NN
run(GEMM_NT, 4096, 4096, 4096, 1, F_BF16X3)   # big x3 (128 tile)
run(GEMM_NT, 1664, 2048, 2048, 1, F_BF16X3)   # mid x3 (64 tile, BK 64)
run(GEMM_NT, 4096, 4096, 4096, 1, 0)          # big fp32
torch.cuda.synchronize()
print('done')
