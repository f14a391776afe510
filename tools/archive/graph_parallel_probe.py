import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'd-lsg-video-caption_amd'))
import torch
from dlsg_amd.hip import HipOps
ops = HipOps()
dev = 'cuda'
B, H = 64, 1024
def mk():
    g = torch.Generator(device=dev).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)
    return dict(slabs=r(4, B, 4*H), bi=r(4*H), bh=r(4*H), cp=r(B, H), c=r(B, H), hd=r(B, H), gates=r(B, 4*H), gl=r(H), bl=r(H), dout=r(B, H), stl=r(B, 2))
a, b = mk(), mk()
def tail(t):
    ops.dec_tail_fwd(t['slabs'], t['bi'], t['bh'], t['cp'], t['c'], t['hd'], t['gates'], (t['gl'], t['bl']), t['dout'], t['stl'], 0.0, 1, seed=0)
N = 200
def capture(par):
    side = torch.cuda.Stream()
    s2 = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        tail(a); tail(b); side.synchronize()
        g = torch.cuda.CUDAGraph()
        g.capture_begin(capture_error_mode='thread_local')
        if par:
            s2.wait_stream(side)
            for _ in range(N): tail(a)
            with torch.cuda.stream(s2):
                for _ in range(N): tail(b)
            side.wait_stream(s2)
        else:
            for _ in range(N): tail(a)
            for _ in range(N): tail(b)
        g.capture_end()
    torch.cuda.current_stream().wait_stream(side)
    return g
def capture_fj():
    side = torch.cuda.Stream()
    s2 = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        g.capture_begin(capture_error_mode='thread_local')
        for _ in range(N):
            s2.wait_stream(side)
            tail(a)
            with torch.cuda.stream(s2):
                tail(b)
            side.wait_stream(s2)
        g.capture_end()
    torch.cuda.current_stream().wait_stream(side)
    return g
g = capture_fj()
g.replay(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): g.replay()
torch.cuda.synchronize()
print('fork/join per pair', (time.perf_counter() - t0) / 5 * 1e3, 'ms for', 2 * N, 'kernels')
for par in (False, True):
    g = capture(par)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    print('parallel' if par else 'serial', (time.perf_counter() - t0) / 5 * 1e3, 'ms for', 2 * N, 'kernels')
