"""Target for `rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace`: long fp32-MFMA GEMM dispatches whose effective shader clock is
GRBM_GUI_ACTIVE / 8 / duration (MI355X_MICROARCH.md, DVFS give-back: the counter sums the 8 XCDs; trustworthy on dispatches of
several ms).  8192^3 (1.1 TFLOP, ~10 ms) and the step's region projection (26624 x 1024 x 2048, two groups, ~2 ms)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
import torch  # noqa: E402
from dlsg_amd.hip import HipOps, GEMM_NT, F_TANH  # noqa: E402

ops = HipOps()
A = torch.randn(8192, 8192, device='cuda'); B = torch.randn(8192, 8192, device='cuda'); C = torch.empty(8192, 8192, device='cuda')
for _ in range(6):
    ops.gemm(GEMM_NT, [(A, B, C)])
R = torch.randn(26624, 2048, device='cuda'); W = torch.randn(1024, 2048, device='cuda'); W2 = torch.randn(1024, 2048, device='cuda')
b = torch.randn(1024, device='cuda'); Y = torch.empty(26624, 1024, device='cuda'); Y2 = torch.empty(26624, 1024, device='cuda')
for _ in range(6):
    ops.gemm(GEMM_NT, [(R, W, Y, b), (R, W2, Y2, b)], flags=F_TANH)
torch.cuda.synchronize()
print('done')
