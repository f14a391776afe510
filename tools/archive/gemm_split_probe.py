"""The step's two largest launches in isolation: the region projection of both streams (NT 26624 x 1024 x 2048, two groups, tanh)
and the deep obj_embed weight gradient (TN, 16 row-chunk groups -> slabs), at batch 64 and 128.  Used to measure (and reject) a
row-panel split that gave the 128-tile kernel whole 768-slot rounds and the rest to the 64-tile kernel: 1.857 vs 1.832 ms and
1.813 vs 1.759 ms at batch 64."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
import torch  # noqa: E402
from dlsg_amd.hip import HipOps, GEMM_NT, GEMM_TN, F_TANH  # noqa: E402

ops = HipOps()


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for B in (64, 128):
    M = B * 26 * 16
    R = torch.randn(M, 2048, device='cuda'); W = torch.randn(1024, 2048, device='cuda'); W2 = torch.randn(1024, 2048, device='cuda')
    b = torch.randn(1024, device='cuda'); Y = torch.empty(M, 1024, device='cuda'); Y2 = torch.empty(M, 1024, device='cuda')
    ms = t(lambda: ops.gemm(GEMM_NT, [(R, W, Y, b), (R, W2, Y2, b)], flags=F_TANH))
    print('B=%d region projection NT %d x 1024 x 2048 x2: %.3f ms  %.1f TFLOP/s' % (B, M, ms, 2 * 2.0 * M * 1024 * 2048 / ms / 1e9))
    dY = torch.randn(M, 1024, device='cuda'); dY2 = torch.randn(M, 1024, device='cuda')
    slabs = torch.empty(16, 1024, 2048, device='cuda')
    st = M // 8
    groups = [(d[i * st:(i + 1) * st], R[i * st:(i + 1) * st], slabs[j * 8 + i]) for j, d in enumerate((dY, dY2)) for i in range(8)]
    ms = t(lambda: ops.gemm(GEMM_TN, groups))
    print('B=%d deep weight gradient TN 1024 x 2048 x %d x2 (16 groups): %.3f ms  %.1f TFLOP/s' % (B, M, ms, 2 * 2.0 * M * 1024 * 2048 / ms / 1e9))
    del R, Y, Y2, dY, dY2
