"""Times the fused decoder-step kernels in isolation (HIP events, MSVD-shaped sizes).
usage: python3 tools/archive/decstep_bench.py [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
import torch  # noqa: E402
from dlsg_amd.hip import HipOps  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
Q, H, D, P, ns, S = 1024, 1024, 1024, 8, 2, 5
ops = HipOps()
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(0)


def r(*shape):
    return torch.randn(*shape, device=dev, generator=g)


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


t = dict(slabs=r(S, B, 4 * Q), add=r(B, 4 * Q), bi=r(4 * Q), bh=r(4 * Q), cp=r(B, Q), c=r(B, Q), h=r(B, Q), gates=r(B, 4 * Q),
         gq=r(Q), bq=r(Q), qcur=r(B, Q), stq=r(B, 2), alpha=r(B, ns * P), slabs2=r(4, B, 4 * D), bi2=r(4 * D), bh2=r(4 * D),
         cp2=r(B, D), c2=r(B, D), hd=r(B, D), gates2=r(B, 4 * D), gl=r(D), bl=r(D), dout=r(B, D), stl=r(B, 2))
K = [r(B, P, Q) * 0.2 for _ in range(ns)]
V = [r(B, P, H) for _ in range(ns)]
ln = [(r(H), r(H)) for _ in range(ns)]
cpre = [r(B, H) for _ in range(ns)]
ctx = [r(B, H) for _ in range(ns)]
stc = [r(B, 2) for _ in range(ns)]


def mid():
    ops.dec_mid_fwd(t['slabs'], t['add'], t['bi'], t['bh'], t['cp'], t['c'], t['h'], t['gates'], (t['gq'], t['bq']), t['qcur'],
                    t['stq'], 0.3, 11, K, V, ln, cpre, ctx, stc, t['alpha'], [0.2, 0.4], [21, 22], 0.03, seed=5)


def tail():
    ops.dec_tail_fwd(t['slabs2'], t['bi2'], t['bh2'], t['cp2'], t['c2'], t['hd'], t['gates2'], (t['gl'], t['bl']), t['dout'],
                     t['stl'], 0.3, 31, seed=5)


def unfused():
    ops.lstm_pw_fwd(t['slabs'], t['c'], B, Q, addend=t['add'], b_ih=t['bi'], b_hh=t['bh'], c_prev=t['cp'], h=t['h'],
                    gates=t['gates'])
    ops.rowln_fwd(t['h'], t['gq'], t['bq'], t['qcur'], t['stq'], p1=0.3, site1=11, seed=5)
    ops.decatt_fwd(K, V, t['qcur'], cpre, t['alpha'], 0.03)
    for i in range(ns):
        ops.rowln_fwd(cpre[i], ln[i][0], ln[i][1], ctx[i], stc[i], pre_tanh=1, p1=0.2, site1=21 + i, seed=5)


print('B=%d  dec_mid_fwd %.1f us   dec_tail_fwd %.1f us   unfused mid chain (5 launches) %.1f us' %
      (B, timeit(mid), timeit(tail), timeit(unfused)))
for name in ('dec_mid_bwd', 'dec_tail_bwd'):
    if hasattr(ops, name + '_bench'):
        print(name, getattr(ops, name + '_bench')(B))
