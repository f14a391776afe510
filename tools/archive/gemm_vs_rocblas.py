"""The step's largest products on this repo's fp32 GEMM against PyTorch-ROCm's library call (rocBLAS / hipBLASLt, TF32 off) on the
same operands: a ceiling check, not a product path.  usage: python3 tools/archive/gemm_vs_rocblas.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
import torch  # noqa: E402
from dlsg_amd.hip import HipOps, GEMM_NT, GEMM_NN, GEMM_TN, F_TILE256, F_FORCE128  # noqa: E402

torch.backends.cuda.matmul.allow_tf32 = False
ops = HipOps()


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(gr, stream=st):
            for _ in range(reps):
                fn()
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3


out = {}
CASES = (('NT region projection 26624 x 1024 x 2048', GEMM_NT, 26624, 1024, 2048),
         ('NN input gradient 26624 x 2048 x 1024', GEMM_NN, 26624, 2048, 1024),
         ('TN weight gradient 1024 x 2048 x 26624', GEMM_TN, 1024, 2048, 26624),
         ('NT 8192^3', GEMM_NT, 8192, 8192, 8192),
         ('NT 1664 x 4096 x 1024', GEMM_NT, 1664, 4096, 1024),
         # mid-size products of the step (one group of the grouped launches)
         ('NT 1664 x 2048 x 2048', GEMM_NT, 1664, 2048, 2048), ('NN 1664 x 2048 x 2048', GEMM_NN, 1664, 2048, 2048),
         ('TN 2048 x 2048 x 1664', GEMM_TN, 2048, 2048, 1664), ('NN 1664 x 1024 x 4096', GEMM_NN, 1664, 1024, 4096),
         ('NT 1664 x 1024 x 6144', GEMM_NT, 1664, 1024, 6144), ('TN 1024 x 6144 x 1664', GEMM_TN, 1024, 6144, 1664),
         ('TN 4096 x 1024 x 1664', GEMM_TN, 4096, 1024, 1664), ('NT 1664 x 1000 x 1024', GEMM_NT, 1664, 1000, 1024),
         # recurrent products (one launch of the word loop as ONE product over the concatenated K)
         ('NT 64 x 4096 x 4096 (language gates)', GEMM_NT, 64, 4096, 4096), ('NT 64 x 4096 x 2348 (query gates)', GEMM_NT, 64, 4096, 2348),
         ('NN 64 x 4096 x 4096 (d language-cell inputs)', GEMM_NN, 64, 4096, 4096), ('NT 128 x 4096 x 4096', GEMM_NT, 128, 4096, 4096),
         # critic
         ('NT 4992 x 2048 x 512', GEMM_NT, 4992, 2048, 512), ('NT 1664 x 512 x 512', GEMM_NT, 1664, 512, 512),
         ('TN 512 x 1536 x 6656', GEMM_TN, 512, 1536, 6656), ('NN 4992 x 512 x 1536', GEMM_NN, 4992, 512, 1536))
for name, mode, M, N, K in CASES:
    if mode == GEMM_NT:
        A, B = torch.randn(M, K, device='cuda'), torch.randn(N, K, device='cuda')
        lib = lambda: torch.mm(A, B.t(), out=C2)
    elif mode == GEMM_NN:
        A, B = torch.randn(M, K, device='cuda'), torch.randn(K, N, device='cuda')
        lib = lambda: torch.mm(A, B, out=C2)
    else:
        A, B = torch.randn(K, M, device='cuda'), torch.randn(K, N, device='cuda')
        lib = lambda: torch.mm(A.t(), B, out=C2)
    C1, C2 = torch.empty(M, N, device='cuda'), torch.empty(M, N, device='cuda')
    us_lib = timeit(lib)
    gf = 2.0 * M * N * K / 1e9
    row = {'library_us': round(us_lib, 1), 'library_TFLOPs': round(gf / us_lib * 1e3, 1)}
    for tag, fl in (('dispatcher', 0),) + ((('tile 256x256', F_TILE256), ('tile 256x128', F_TILE256 | F_FORCE128)) if (K % 4 == 0 and M >= 1024) else ()):
        C1.zero_()
        us_own = timeit(lambda: ops.gemm(mode, [(A, B, C1)], flags=fl))
        err = ((C1 - C2).abs().max() / C2.abs().max()).item()
        assert err < 2e-5, (name, tag, err)
        row[tag] = {'us': round(us_own, 1), 'TFLOPs': round(gf / us_own * 1e3, 1), 'max_rel_diff_vs_library': err}
    out[name] = row
print(json.dumps(out, indent=1))
