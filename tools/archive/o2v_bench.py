"""GPU microbenchmark of the fused object->frame graph kernel (dlsg_o2v_fwd) in isolation.
bytes = 4*B*(T*O*H + 2*T*H) (read y once, read v, write z)."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
import torch  # noqa: E402
from dlsg_amd.hip import HipOps  # noqa: E402

ops = HipOps()
T, H = 26, 1024
for O in (16, 36):
    NO = T * O
    for B in (64, 128, 256, 512):
        y = torch.tanh(torch.randn(B, NO, H, device='cuda'))
        v = torch.randn(B, T, H, device='cuda')
        g, b_ = torch.ones(H, device='cuda'), torch.zeros(H, device='cuda')
        z = torch.empty(B * T, H, device='cuda'); ml = torch.empty(B * T, 2, device='cuda')
        st = torch.empty(B * NO, 2, device='cuda'); S = torch.empty(B, NO, T, device='cuda')
        tiles = (NO + 31) // 32
        res = []
        for ns in sorted(set([1, 2, 4, max(1, min(tiles, 256 // B)), max(1, min(tiles, 512 // B))])):
            for _ in range(2):
                ops.o2v_fwd(y, v, g, b_, z, ml, st, S, 1 / math.sqrt(2048), ns)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.o2v_fwd(y, v, g, b_, z, ml, st, S, 1 / math.sqrt(2048), ns)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            gb = 4.0 * B * (NO * H + 2 * T * H) / 1e9
            res.append('ns%d: %.3f ms %.0f GB/s (%.1f%%)' % (ns, ms, gb / ms * 1e3, gb / ms * 1e3 / 80))
        print('O=%d B=%d  %s' % (O, B, '  '.join(res)))
        # backward (scores pass + apply pass); bytes = y read once + dy written + dz, v read + dv written
        dz = torch.randn(B, T, H, device='cuda'); dy = torch.empty(B, NO, H, device='cuda')
        dv = torch.empty(B, T, H, device='cuda')
        ns = max(1, min(tiles, 256 // B))
        for _ in range(2):
            ops.o2v_bwd(y, st, g, b_, v, z.view(B, T, H), dz, S, ml, dy, dv, 1 / math.sqrt(2048), ns)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.o2v_bwd(y, st, g, b_, v, z.view(B, T, H), dz, S, ml, dy, dv, 1 / math.sqrt(2048), ns)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        gb = 4.0 * B * (2 * NO * H + 3 * T * H) / 1e9
        print('          backward ns%d: %.3f ms %.0f GB/s (%.1f%%)' % (ns, ms, gb / ms * 1e3, gb / ms * 1e3 / 80))
        del y, v, z, S, dz, dy, dv
