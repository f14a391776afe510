"""The BiLSTM recurrence of EncoderVisual in isolation (batch 64, 26 steps, H = 1024 by default): the persistent launch
(csrc/bilstm.hip) against the per-step schedule (grouped skinny GEMM + pointwise launch per step), eager and graph-replayed.
usage: python3 tools/bilstm_bench.py [B] [T] [H]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
import torch  # noqa: E402
from dlsg_amd.hip import HipOps  # noqa: E402
from dlsg_amd import engine as E  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 26
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
ops = HipOps()
g = torch.Generator().manual_seed(0)
dev = 'cuda'
xg = [torch.randn(B * T, 4 * H, generator=g).to(dev) for _ in range(2)]
Whh = [(torch.randn(4 * H, H, generator=g) / H ** 0.5).to(dev) for _ in range(2)]
bih = [torch.randn(4 * H, generator=g).to(dev) * 0.1 for _ in range(2)]
bhh = [torch.randn(4 * H, generator=g).to(dev) * 0.1 for _ in range(2)]
out = torch.empty(B, T, 2 * H, device=dev)
hprev = [torch.zeros(B, T, H, device=dev) for _ in range(2)]
cst = [torch.empty(B, T, H, device=dev) for _ in range(2)]
gates = [torch.empty(B, T, 4 * H, device=dev) for _ in range(2)]


def chunks():
    return [(b0, min(B, b0 + 64)) for b0 in range(0, B, 64)]


def persistent():
    for b0, b1 in chunks():          # the kernel takes <= 64 rows: larger batches run it once per 64-row chunk
        ops.bilstm_fwd([x.view(B, T, 4 * H)[b0:b1].view(-1, 4 * H) for x in xg], Whh, bih, bhh, out[b0:b1], [h[b0:b1] for h in hprev],
                       [c[b0:b1] for c in cst], [g_[b0:b1] for g_ in gates])


def steps():
    E._bilstm_steps_fwd(ops, xg, Whh, bih, bhh, out, hprev, cst, gates, B, T, H, out)


def timeit(fn, n=20, graph=False):
    fn(); fn()
    torch.cuda.synchronize()
    if graph:
        gr = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn()                                # (per-stream scratch -- the stream-K workspace -- must exist before the capture)
            gr.capture_begin()
            fn()
            gr.capture_end()
        torch.cuda.current_stream().wait_stream(s)
        run = gr.replay
    else:
        run = fn
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


dout = torch.randn(B, T, 2 * H, generator=g).to(dev)
dG = [torch.empty(B, T, 4 * H, device=dev) for _ in range(2)]


def persistent_bwd():
    for b0, b1 in chunks():
        ops.bilstm_bwd([g_[b0:b1] for g_ in gates], [c[b0:b1] for c in cst], dout[b0:b1], Whh, [d_[b0:b1] for d_ in dG])


def steps_bwd():
    E._bilstm_steps_bwd(ops, gates, cst, dout, Whh, dG, B, T, H, out)


persistent()
torch.cuda.synchronize()
ref = out.clone()
steps()
torch.cuda.synchronize()
print('max |persistent - per-step| on h: %.3g   time-out word: %d' % ((ref - out).abs().max().item(), int(ops._bilstm_err.item())))
persistent_bwd()
torch.cuda.synchronize()
refg = dG[0].clone()
steps_bwd()
torch.cuda.synchronize()
print('max |persistent - per-step| on dG: %.3g (scale %.3g)  time-out word: %d' % ((refg - dG[0]).abs().max().item(), refg.abs().max().item(),
                                                                               int(ops._bilstm_err.item())))
for name, fn in (('persistent', persistent), ('per-step', steps), ('persist-bwd', persistent_bwd), ('perstep-bwd', steps_bwd)):
    for graph in (False, True):
        ms = timeit(fn, graph=graph)
        print('%-11s %-6s B=%d T=%d H=%d: %.3f ms per sequence = %.2f us per step' % (name, 'graph' if graph else 'eager', B, T, H, ms,
                                                                                   ms * 1e3 / T))
