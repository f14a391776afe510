"""Does the stream-K launch run slower from cold caches?  The region projection (NT 26624 x 1024 x 2048 x 2, bias + tanh) and the deep
weight gradient (TN 1024 x 2048 x 26624 x 2) timed (a) back to back, (b) each behind a 1-GB fill that evicts L2 / Infinity Cache,
(c) behind a burst of small kernels.  HIP events around single launches.  usage: python3 tools/archive/gemm_sk_cold_probe.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
import torch  # noqa: E402
from dlsg_amd.hip import HipOps, GEMM_NT, GEMM_TN, F_SK, F_TANH  # noqa: E402

ops = HipOps()
g = torch.Generator(device='cuda')
g.manual_seed(1)
A = torch.randn(26624, 2048, device='cuda', generator=g)
Bs = [torch.randn(1024, 2048, device='cuda', generator=g) for _ in range(2)]
bias = [torch.randn(1024, device='cuda', generator=g) for _ in range(2)]
Cs = [torch.empty(26624, 1024, device='cuda') for _ in range(2)]
dy = [torch.randn(26624, 1024, device='cuda', generator=g) for _ in range(2)]
G = [torch.zeros(1024, 2048, device='cuda') for _ in range(2)]
junk = torch.empty(256 * 1024 * 1024, device='cuda')


def nt():
    ops.gemm(GEMM_NT, [(A, Bs[i], Cs[i], bias[i]) for i in range(2)], flags=F_TANH | F_SK)


def tn():
    ops.gemm(GEMM_TN, [(dy[i], A, G[i]) for i in range(2)], flags=F_SK)


def timed(fn, before, n=12):
    ts = []
    for _ in range(n):
        before()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return round(ts[len(ts) // 2], 1)


def evict():
    junk.fill_(1.0)


def small_kernels():
    x = junk[:1 << 20]
    for _ in range(200):
        x.add_(1.0)


for _ in range(20):
    nt(); tn()
torch.cuda.synchronize()
out = {}
for name, fn in (('region projection NT', nt), ('deep weight gradient TN', tn)):
    out[name] = {'back_to_back_us': timed(fn, lambda: None), 'behind_a_1GB_fill_us': timed(fn, evict),
                 'behind_200_small_kernels_us': timed(fn, small_kernels)}
print(json.dumps(out, indent=1))
