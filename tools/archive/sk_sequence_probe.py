"""What does the launch in front of a stream-K GEMM do to its duration?  Sequences of launches, each sequence repeated back to back
(asynchronously queued, the GPU never idles inside a sequence), run under `rocprofv3 --kernel-trace`; the per-dispatch durations of
the region-projection launch are then grouped by sequence (tools/archive/sk_sequence_report.py reads the kernel-trace CSV).
  A: NT NT NT ...                              (back to back)
  B: [0.5-ms memory-bound update, NT]          (what Adam -> region projection looks like in the replayed step)
  C: [100 x 20-us light kernels, NT]           (a word loop's worth of light launches in front)
  D: [TN deep, NT]                             (heavy matrix kernel in front)
  E: [idle 2 ms (host sleep), NT]              (the stream runs dry)
usage: rocprofv3 --kernel-trace --output-format csv -d <out> -- python3 tools/archive/sk_sequence_probe.py"""
import os
import sys
import time

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
big = torch.zeros(90 * 1024 * 1024, device='cuda')          # 360 MB: an Adam-sized stream
small = torch.zeros(1 << 20, device='cuda')
marker = torch.zeros(1234, device='cuda')                   # a fill of this size separates the sequences in the trace


def nt():
    ops.gemm(GEMM_NT, [(A, Bs[i], Cs[i], bias[i]) for i in range(2)], flags=F_TANH | F_SK)


def tn():
    ops.gemm(GEMM_TN, [(dy[i], A, G[i]) for i in range(2)], flags=F_SK)


def mark(k):
    for _ in range(k):
        marker.fill_(1.0)


for _ in range(30):
    nt(); tn()
torch.cuda.synchronize()
REP = 12
mark(1)
for _ in range(REP):
    nt()
torch.cuda.synchronize(); mark(2)
for _ in range(REP):
    big.mul_(1.0001); big.add_(1.0); nt()
torch.cuda.synchronize(); mark(3)
for _ in range(REP):
    for _ in range(100):
        small.add_(1.0)
    nt()
torch.cuda.synchronize(); mark(4)
for _ in range(REP):
    tn(); nt()
torch.cuda.synchronize(); mark(5)
for _ in range(REP):
    torch.cuda.synchronize(); time.sleep(0.002); nt()
torch.cuda.synchronize(); mark(6)
print('done')
