"""The recurrent gate products of a word step (M = batch rows) on the (32 MI) x 128 shared-ring kernel (csrc/gemm.hip, generation 4,
flag F_RING128) against the dispatcher's choice (generations 1-2), per K split.  usage: python3 tools/skinny4_probe.py [batch=64] [experiment library]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
import torch  # noqa: E402
import dlsg_amd.hip as _H  # noqa: E402
from dlsg_amd.hip import HipOps, GEMM_NT, GEMM_NN, F_RING128  # noqa: E402

EXP = len(sys.argv) > 2            # an experiment build of the library (results are not checked)
if EXP:
    _H.load_library.__defaults__ = (os.path.abspath(sys.argv[2]),)

ops = HipOps()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.Generator(device='cuda').manual_seed(0)
out = {}


def r(*s):
    return torch.randn(*s, device='cuda', generator=g)


def timeit(fn, reps=40):
    """device time per launch: the launches are captured into one hipGraph (host-side descriptor marshalling of a 16-group
    launch takes longer than the kernel)"""
    fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(gr, stream=st):
            for _ in range(reps):
                fn()
    for _ in range(2):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / (3 * reps) * 1e3, 1)


def split(k, chunk):
    n = max(1, (k + chunk - 1) // chunk)
    step = ((k + n - 1) // n + 31) // 32 * 32
    return [(k0, min(k, k0 + step)) for k0 in range(0, k, step)]


def nt_case(name, N, segs):
    W = r(N, sum(segs))
    xs = [r(B, k) for k in segs]
    ref = torch.cat(xs, 1).double() @ W.double().t()
    row = {}
    for chunk in (1024, 512, 352, 256):
        pieces, c0 = [], 0
        for x, k in zip(xs, segs):
            for k0, k1 in split(k, chunk):
                pieces.append((x[:, k0:k1], W[:, c0 + k0:c0 + k1]))
            c0 += k
        if len(pieces) > 16:
            continue
        slabs = torch.empty(len(pieces), B, N, device='cuda')
        groups = [(a_, b_, slabs[i]) for i, (a_, b_) in enumerate(pieces)]
        for tag, fl in (('old', 0), ('ring128', F_RING128)):
            slabs.zero_()
            us = timeit(lambda: ops.gemm(GEMM_NT, groups, flags=fl))
            err = ((slabs.double().sum(0) - ref).abs().max() / ref.abs().max()).item()
            assert EXP or err < 1e-5, (name, chunk, tag, err)
            row['chunk %d (%d groups) %s' % (chunk, len(pieces), tag)] = us
            if EXP and fl:
                cyc, ticks = slabs[0, 0, 0].item(), slabs[0, 0, 1].item()
                row['chunk %d loop clock GHz / cycles per stage' % chunk] = (round(cyc / max(ticks, 1) * 0.1, 3), round(cyc / ((pieces[0][0].shape[1] + 31) // 32), 1))
    out[name] = row


def nn_case(name, Kc, widths):
    dy = r(B, Kc)
    Ws = [r(Kc, wd) for wd in widths]
    tot = sum(widths)
    ref = dy.double() @ torch.cat(Ws, 1).double()
    row = {}
    for chunk in (1024, 512, 256):
        bounds = split(Kc, chunk)
        if len(bounds) * len(Ws) > 16:
            continue
        slabs = torch.empty(len(bounds), B, tot, device='cuda')
        groups, c0 = [], 0
        for Wm, wd in zip(Ws, widths):
            for i, (k0, k1) in enumerate(bounds):
                groups.append((dy[:, k0:k1], Wm[k0:k1, :], slabs[i][:, c0:c0 + wd]))
            c0 += wd
        for tag, fl in (('old', 0), ('ring128', F_RING128)):
            slabs.zero_()
            us = timeit(lambda: ops.gemm(GEMM_NN, groups, flags=fl))
            err = ((slabs.double().sum(0) - ref).abs().max() / ref.abs().max()).item()
            assert EXP or err < 1e-5, (name, chunk, tag, err)
            row['chunk %d (%d groups) %s' % (chunk, len(groups), tag)] = us
    out[name] = row


nt_case('query gates NT 4096 x [300|1024|1024]', 4096, [300, 1024, 1024])
nt_case('lang gates NT 4096 x [1024 x 4]', 4096, [1024, 1024, 1024, 1024])
nn_case('d lang-cell inputs NN 4096 -> [3072|1024]', 4096, [3072, 1024])
nn_case('d query-cell rec NN 4096 -> [1024|1024]', 4096, [1024, 1024])
nn_case('one weight NN 4096 -> [4096]', 4096, [4096])
print(json.dumps(out, indent=1))
