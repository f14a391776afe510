"""The stream-K kernel's tile choices on the step's mid-size products (hipGraph-replayed launches, back to back): the tiled
dispatcher, 256 x 256, 128 x 256, 128 x 128 stream-K tiles and what dlsg_gemm picks by itself (csrc/gemm_sk.hip, sk_pick_tile /
dlsg_gemm_sk_wanted).  usage: python3 tools/gemm_sk_tiles.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
import torch
from dlsg_amd.hip import HipOps, GEMM_NT, GEMM_NN, GEMM_TN, F_NOSK, F_SK, F_SK_BM128, F_SK_BM256, F_SK_BN128
ops = HipOps()
g = torch.Generator(device='cuda'); g.manual_seed(1)
def timeit(fn, reps=20):
    st = torch.cuda.Stream()
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        fn()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(reps): fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3
def ops_(mode, M, N, K):
    if mode == GEMM_NT: return torch.randn(M, K, device='cuda', generator=g), torch.randn(N, K, device='cuda', generator=g)
    if mode == GEMM_NN: return torch.randn(M, K, device='cuda', generator=g), torch.randn(K, N, device='cuda', generator=g)
    return torch.randn(K, M, device='cuda', generator=g), torch.randn(K, N, device='cuda', generator=g)
CASES = [('TN 4096x{1024x10,300}x1664', GEMM_TN, 4096, [(1024, 1664)] * 10 + [(300, 1664)]), ('NT 1664x10000x1536 (MSR-VTT vocabulary)', GEMM_NT, 1664, [(10000, 1536)]), ('TN 10000x1536x1664', GEMM_TN, 10000, [(1536, 1664)]),
         ('NT 1664x1024x2048', GEMM_NT, 1664, [(1024, 2048)]), ('TN 1024x2048x1664', GEMM_TN, 1024, [(2048, 1664)]),
         ('NT 1664x1000x1024', GEMM_NT, 1664, [(1000, 1024)]), ('TN 1000x1024x1664', GEMM_TN, 1000, [(1024, 1664)]),
         ('NN 1664x2048x1024', GEMM_NN, 1664, [(2048, 1024)]), ('NT 1664x1024x6144', GEMM_NT, 1664, [(1024, 6144)]),
         ('NT 1664x4096x1024 x2', GEMM_NT, 1664, [(4096, 1024)] * 2), ('NN 1664x2048x2048 x3', GEMM_NN, 1664, [(2048, 2048)] * 3),
         ('NN 1664x1024x2048 x4', GEMM_NN, 1664, [(1024, 2048)] * 4), ('NT 1664x2048x2048 x3 + 1024', GEMM_NT, 1664, [(2048, 2048)] * 3 + [(1024, 2048)]),
         ('TN 2048x2048x1664 x3', GEMM_TN, 2048, [(2048, 1664)] * 3), ('TN 1024x{2048,2048,6144}x1664', GEMM_TN, 1024, [(2048, 1664), (2048, 1664), (6144, 1664)]),
         ('NT 320x1024x1024 x4', GEMM_NT, 320, [(1024, 1024)] * 4), ('NN 320x1024x1024 x2', GEMM_NN, 320, [(1024, 1024)] * 2),
         ('TN 1024x1024x320 x8', GEMM_TN, 1024, [(1024, 320)] * 8), ('NN 1664x300x1024 x4', GEMM_NN, 1664, [(300, 1024)] * 4)]
for name, mode, M, nk in CASES:
    groups = []
    for N, K in nk:
        A, B = ops_(mode, M, N, K)
        groups.append((A, B, torch.zeros(M, N, device='cuda')))
    gf = sum(2.0 * M * N * K for N, K in nk) / 1e9
    row = []
    for tag, fl in (('tiled', F_NOSK), ('sk256', F_SK | F_SK_BM256), ('sk128x256', F_SK | F_SK_BM128), ('sk128x128', F_SK | F_SK_BN128), ('default', 0)):
        try:
            us = timeit(lambda: ops.gemm(mode, groups, flags=fl))
            row.append('%s %6.1f us %5.1f TF' % (tag, us, gf / us * 1e3))
        except RuntimeError as e:
            row.append('%s failed' % tag)
    print('%-34s %s' % (name, ' | '.join(row)), flush=True)
