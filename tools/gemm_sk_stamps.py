"""In-kernel timeline of the stream-K GEMM (a `make PROBES=1` build of csrc/gemm_sk.hip): 100-MHz stamps of workgroups v < 8 at the
start, after the last stage of every item, after its store, at the end of the stream and around the fix-up.
usage: python3 tools/gemm_sk_stamps.py [mid]      (mid: the step's mid-size products, where the launch's fixed costs show)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
import torch  # noqa: E402
from dlsg_amd.hip import HipOps, GEMM_NT, GEMM_NN, GEMM_TN, F_SK, F_ACCUM, F_TANH, F_SK_BM128, F_SK_BM256  # noqa: E402

ops = HipOps()
dev = torch.device('cuda', 0)
g = torch.Generator(device='cuda')
g.manual_seed(3)


def run(name, mode, M, nk, fl, use_bias, share_a=False):
    groups, A0 = [], None
    for N, K in nk:
        if mode == GEMM_NT:
            A, B = torch.randn(M, K, device='cuda', generator=g), torch.randn(N, K, device='cuda', generator=g)
        elif mode == GEMM_NN:
            A, B = torch.randn(M, K, device='cuda', generator=g), torch.randn(K, N, device='cuda', generator=g)
        else:
            A, B = torch.randn(K, M, device='cuda', generator=g), torch.randn(K, N, device='cuda', generator=g)
        if share_a:
            A0 = A if A0 is None else A0
            A = A0
        groups.append((A, B, torch.zeros(M, N, device='cuda'), torch.randn(N, device='cuda', generator=g) if use_bias else None))
    for _ in range(30):                      # (the chip raises its clock over the first dispatches)
        ops.gemm(mode, groups, flags=fl | F_SK)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.gemm(mode, groups, flags=fl | F_SK)
    e1.record()
    torch.cuda.synchronize()
    ws = ops._gemm_workspace(dev)
    st = ws[1024:1536].view(torch.int64).view(8, 32).cpu()
    per_stage = ws[2048:4096].view(torch.int64).cpu().tolist()
    rows = []
    for v in range(8):
        t = st[v].tolist()
        t0 = t[0]
        ev = [(i, (x - t0) / 100.0) for i, x in enumerate(t) if x >= t0 and x - t0 < 10 ** 9]
        rows.append(ev)
    print(name, 'launch %.1f us' % (e0.elapsed_time(e1) * 1e3))
    for v, ev in enumerate(rows[:4]):
        items = [(i, round(us, 1)) for i, us in ev]
        print('  v=%d' % v, items)
    # per-item compute and store durations of v = 0
    ev = dict(rows[0])
    seq = []
    prev = 0.0
    i = 1
    while i in ev and i + 1 in ev and i < 27:
        seq.append({'stages_us': round(ev[i] - prev, 1), 'store_us': round(ev[i + 1] - ev[i], 1)})
        prev = ev[i + 1]
        i += 2
    tail = {k: round(ev[k] - prev, 1) for k in (27, 28, 29, 30, 31) if k in ev}
    print('  v=0 items:', json.dumps(seq), 'tail (27 stream end; 28 / 29 decided / added up for the first split item, 30 / 31 the second):', tail, flush=True)
    ts = [x for x in per_stage if x > 0]
    d = [(b - a) / 100.0 for a, b in zip(ts, ts[1:])]
    if d:
        print('  v=0 us per stage (first 80):', ' '.join('%.1f' % x for x in d[:80]))
        srt = sorted(d)
        print('  v=0 stage time: median %.2f, p10 %.2f, p90 %.2f, n=%d' % (srt[len(srt) // 2], srt[len(srt) // 10], srt[9 * len(srt) // 10], len(d)), flush=True)
    ws[1024:4096].zero_()


if len(sys.argv) > 1 and sys.argv[1] == 'mid':
    run('NT 1664x1024x2048 bm128', GEMM_NT, 1664, [(1024, 2048)], F_SK_BM128, True)
    run('NT 1664x1024x2048 bm256', GEMM_NT, 1664, [(1024, 2048)], F_SK_BM256, True)
    run('TN 1024x2048x1664 bm128', GEMM_TN, 1024, [(2048, 1664)], F_SK_BM128, False)
    run('NT 1664x1000x1024 bm128', GEMM_NT, 1664, [(1000, 1024)], F_SK_BM128, True)
    run('NN 1664x2048x2048 x3 bm128', GEMM_NN, 1664, [(2048, 2048)] * 3, F_SK_BM128, False)
    sys.exit(0)
run('region NT 26624x1024x2048 x2 bias tanh', GEMM_NT, 26624, [(1024, 2048)] * 2, F_TANH, True, True)
run('region NT 26624x1024x2048 x2 plain', GEMM_NT, 26624, [(1024, 2048)] * 2, 0, False, True)
run('TN 1024x2048x26624 x2 accum', GEMM_TN, 1024, [(2048, 26624)] * 2, F_ACCUM, False)
run('NT 8192^3', GEMM_NT, 8192, [(8192, 8192)], 0, False)
run('NT 1024x2048x26624 x2 (deep NT)', GEMM_NT, 1024, [(2048, 26624)] * 2, 0, False)
run('TN 4096x1024x1664 x7 accum', GEMM_TN, 4096, [(1024, 1664)] * 7, F_ACCUM, False)
