"""Wave quantisation of the step's two largest launches (batch 64, MSVD-shaped; `msrvtt` as argv[1] for 36 regions):
(a) region projection of both streams, NT 26624 x 1024 x 2048 x 2 = 3328 tiles = 4.33 rounds of the 768 workgroup slots:
    one launch vs main rows (whole rounds) + K-split tail (engine.region_projections, TAIL_SPLIT);
(b) deep obj_embed weight gradient, TN 1024 x 2048 x 26624 per stream, row chunks -> slabs: 8 chunks x 2 streams in one launch
    (2048 workgroups = 2.67 rounds) vs 6 (1536 = 2 rounds) vs 12 chunks per stream, one launch per stream (2 x 1536) vs 4.
usage: python3 tools/gemm_rounds_probe.py [msvd|msrvtt] [batch]"""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
import torch  # noqa: E402
from dlsg_amd.hip import HipOps  # noqa: E402
from dlsg_amd import engine as E  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else 'msvd'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
O = 16 if shape == 'msvd' else 36
T, R, H = 26, 2048, 1024
ops = HipOps()
g = torch.Generator().manual_seed(0)
regions = torch.randn(B, T, O, R, generator=g).cuda()
mods = []
for i in range(2):
    m = types.SimpleNamespace()
    m.obj_embed = types.SimpleNamespace(weight=(torch.randn(H, R, generator=g) / R ** 0.5).cuda(), bias=torch.randn(H, generator=g).cuda())
    mods.append(m)
rows = B * T * O
dys = [torch.randn(rows, H, generator=g).cuda() for _ in range(2)]
gouts = [torch.zeros(H, R, device='cuda') for _ in range(2)]


def timed(fn, n=8):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


flops_a = 2.0 * rows * H * R * 2
ref = None
for rep in range(2):
    for tail in (False, True):
        E.TAIL_SPLIT = tail
        ms = timed(lambda: E.region_projections(ops, mods, regions))
        ys = E.region_projections(ops, mods, regions)
        torch.cuda.synchronize()
        if ref is None:
            ref = [y.clone() for y in ys]
        err = max((a - b).abs().max().item() for a, b in zip(ref, ys))
        print('region projection tail_split=%-5s plan=%s: %.3f ms = %.1f TFLOP/s   max|diff| %.2g' % (
            tail, E._tail_plan(rows, H, 2) if tail else None, ms, flops_a / ms / 1e9, err))
flops_b = 2.0 * rows * H * R * 2
for rep in range(2):
    for ks in (8, 6, 12, 4):
        E.DEEP_TN_CHUNKS = ks

        def run():
            E.gemm_tn_deep(ops, [(dys[i], regions.view(rows, R), gouts[i]) for i in range(2)], regions)
        ms = timed(run)
        print('deep TN chunks per stream=%2d: %.3f ms = %.1f TFLOP/s' % (ks, ms, flops_b / ms / 1e9))
