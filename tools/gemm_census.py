"""Census of the GEMM launches of one train step (kernel by kernel, batch 64 MSVD-shaped by default): every dlsg_gemm call
bracketed by events, grouped by (mode, M, N, K per group, groups, batch count); count, total time and TFLOP/s per shape.
usage: python tools/gemm_census.py [fp32|x3_bwd|x3_all] [batch] [msvd|msrvtt]"""
import collections
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
import torch  # noqa: E402
import dlsg_amd  # noqa: E402
from dlsg_amd.synth import synth_state_dict, synth_batch  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
shape = sys.argv[3] if len(sys.argv) > 3 else 'msvd'
args = dlsg_amd.msvd_shaped() if shape == 'msvd' else dlsg_amd.msrvtt_shaped()
V = 1000 if shape == 'msvd' else 10000
torch.manual_seed(0)
net = dlsg_amd.CapGnnModel(args, dlsg_amd.make_vocab(V))
net.load_state_dict(synth_state_dict(net.state_dict(), 0))
net = net.cuda().train()
net.gemm_precision = mode
frames, regions, caps, lens = [t.cuda() for t in synth_batch(args, V, B, 1)]
tr = dlsg_amd.Trainer(net, use_graphs=False)
random.seed(12)
ops = net.ops
real = ops.gemm
log = []


def traced(gmode, groups, *a, **kw):
    C0 = groups[0][2]
    nb = C0.size(0) if C0.dim() == 3 else 1
    M = C0.shape[-2]
    Ns = tuple(g[2].shape[-1] for g in groups)
    Ks = tuple((g[0].shape[-2] if gmode == 2 else g[0].shape[-1]) for g in groups)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = real(gmode, groups, *a, **kw)
    e1.record()
    log.append((('nt', 'nn', 'tn')[gmode], M, Ns, Ks, nb, e0, e1))
    return r


for it in range(3):
    if it == 2:
        ops.gemm = traced
    tr.step(frames, regions, caps, lens, dlsg_amd.ss_epsilon(0))
torch.cuda.synchronize()
if os.environ.get('CENSUS_SEQUENCE'):             # the launches in issue order (per-word-step skinny launches folded)
    prev, rep = None, 0
    for rec in log + [None]:
        if rec is None:
            break
        m, M, Ns, Ks, nb, e0, e1 = rec
        key = (m, M, Ns, Ks, nb)
        us = e0.elapsed_time(e1) * 1e3
        if M <= 64 and key == prev:
            rep += 1
            continue
        if rep:
            print('      ... x%d more' % rep)
        rep = 0
        prev = key
        print('%-3s M=%-6d N=%-30s K=%-30s nb=%-3d %8.1f us' % (m, M, str(Ns)[:30], str(Ks)[:30], nb, us))
agg = collections.OrderedDict()
for m, M, Ns, Ks, nb, e0, e1 in log:
    key = (m, M, Ns if len(set(Ns)) > 1 else (Ns[0],), Ks if len(set(Ks)) > 1 else (Ks[0],), len(Ns), nb)
    flops = 2.0 * M * nb * sum(n * k for n, k in zip(Ns, Ks))
    d = agg.setdefault(key, [0, 0.0, 0.0])
    d[0] += 1; d[1] += e0.elapsed_time(e1) * 1e3; d[2] += flops
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
print('%-3s %6s %-22s %-28s %3s %3s %5s %9s %8s %7s' % ('op', 'M', 'N', 'K', 'grp', 'nb', 'calls', 'total us', 'us/call', 'TF/s'))
for (m, M, Ns, Ks, ng, nb), (c, us, fl) in rows:
    print('%-3s %6d %-22s %-28s %3d %3d %5d %9.0f %8.1f %7.1f' % (m, M, str(Ns)[:22], str(Ks)[:28], ng, nb, c, us, us / c, fl / us / 1e6))
print('event-bracketed GEMM time of the step: %.2f ms in %d launches' % (tot / 1e3, len(log)))
