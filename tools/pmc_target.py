"""Target for the rocprofv3 --pmc passes: a few launches of the kernels whose HBM traffic profiles/traffic.json records.
  cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d <out>/fetch -- python3 tools/pmc_target.py
  cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d <out>/write -- python3 tools/pmc_target.py
"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
import torch  # noqa: E402
from dlsg_amd.hip import HipOps, GEMM_NT, GEMM_TN, F_BF16X3, F_TANH  # noqa: E402

ops = HipOps()
dev = 'cuda'
# 1. region projection, batch 64: (B*T*O=26624, 2048) x (1024, 2048)^T, tanh epilogue (layer.py:184)
A = torch.randn(26624, 2048, device=dev); W = torch.randn(1024, 2048, device=dev); b = torch.randn(1024, device=dev)
C = torch.empty(26624, 1024, device=dev)
W2 = torch.randn(1024, 2048, device=dev); b2 = torch.randn(1024, device=dev); C2 = torch.empty(26624, 1024, device=dev)
for _ in range(3):      # as the step launches it: both streams' projections as two groups of one launch
    ops.gemm(GEMM_NT, [(A, W, C, b), (A, W2, C2, b2)], flags=F_TANH)
for _ in range(3):
    ops.gemm(GEMM_NT, [(A, W, C, b), (A, W2, C2, b2)], flags=F_TANH | F_BF16X3)
# 2. its weight gradient (TN, 26624 deep, 8 row groups -> slabs)
dY = torch.randn(26624, 1024, device=dev)
slabs = torch.empty(8, 1024, 2048, device=dev)
step = 26624 // 8
for fl in (0, F_BF16X3):
    for _ in range(3):
        ops.gemm(GEMM_TN, [(dY[i * step:(i + 1) * step], A[i * step:(i + 1) * step], slabs[i]) for i in range(8)], flags=fl)
# 3. fused object->frame graph, 256 clips, MSVD-shaped
B, T, O, H = 256, 26, 16, 1024
NO = T * O
y = torch.tanh(torch.randn(B, NO, H, device=dev)); v = torch.randn(B, T, H, device=dev)
g, be = torch.ones(H, device=dev), torch.zeros(H, device=dev)
z = torch.empty(B * T, H, device=dev); ml = torch.empty(B * T, 2, device=dev)
st = torch.empty(B * NO, 2, device=dev); S = torch.empty(B, NO, T, device=dev)
for _ in range(3):
    ops.o2v_fwd(y, v, g, be, z, ml, st, S, 1 / math.sqrt(2048), 1)
torch.cuda.synchronize()
print('done')
