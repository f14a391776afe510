"""Tile / K-split choices for the products of a critic update (dlsg_amd/critic.py) at batch 64: every (mode, M, N, K) of the schedule
under the dispatcher's default tile, each forced tile, and -- for the deep TN weight gradients -- K split over groups writing slabs
(+ the slab_reduce that folds them).  usage: python3 tools/archive/critic_gemm_probe.py [batch=64]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
import torch  # noqa: E402
from dlsg_amd.hip import HipOps, GEMM_NT, GEMM_NN, GEMM_TN, F_FORCE64, F_FORCE128  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L, T = 26, 3
ops = HipOps()
dev = 'cuda'
TILES = {'default': 0, '64x64': F_FORCE64, '128x128': F_FORCE128, '128x64': F_FORCE64 | F_FORCE128}


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def rnd(*s):
    return torch.randn(*s, device=dev)


out = {}
R3, R1, R4 = 3 * B * L, B * L, 4 * B * L
# forward-like (NT) and input-gradient (NN) products: (rows, N, K)
for mode, name in ((GEMM_NT, 'NT'), (GEMM_NN, 'NN')):
    for M in (R3, R1):
        for N, K in ((512, 1536), (2048, 512), (1536, 512), (512, 512), (512, 2048), (1536, 512)):
            A = rnd(M, K)
            Bm = rnd(N, K) if mode == GEMM_NT else rnd(K, N)
            Cc = rnd(M, N)
            row = {}
            for tname, fl in TILES.items():
                row[tname] = round(timeit(lambda: ops.gemm(mode, [(A, Bm, Cc)], flags=fl)), 1)
            gf = 2.0 * M * N * K / 1e9
            row['GFLOP'] = round(gf, 2)
            row['best_TFLOPs'] = round(gf / min(v for k, v in row.items() if k in TILES) * 1e3, 1)
            out['%s M=%d N=%d K=%d' % (name, M, N, K)] = row
# two heads as two groups
for M in (R3, R1, 3 * B * T):
    A = rnd(M, 512)
    W0, W1, C0, C1 = rnd(512, 512), rnd(512, 512), rnd(M, 512), rnd(M, 512)
    row = {}
    for tname, fl in TILES.items():
        row[tname] = round(timeit(lambda: ops.gemm(GEMM_NT, [(A, W0, C0), (A, W1, C1)], flags=fl)), 1)
    out['NT 2 groups M=%d N=512 K=512' % M] = row
# weight gradients (TN): out (M, N) = A (K, M)^T B (K, N), K = 4 B L rows
for M, N, K in ((512, 1536, R4), (2048, 512, R4), (1536, 512, R4), (512, 512, R4), (512, 512, 4 * B * T), (512, 1000, R1), (512, 512, R1)):
    A, Bm, Cc = rnd(K, M), rnd(K, N), rnd(M, N)
    row = {}
    for ns in (1, 2, 4, 8, 13, 16):
        if K // ns < 128:
            continue
        step = ((K + ns - 1) // ns + 31) // 32 * 32
        bounds = [(k, min(K, k + step)) for k in range(0, K, step)]
        slabs = rnd(len(bounds), M, N)
        for tname, fl in TILES.items():
            def run():
                if len(bounds) == 1:
                    ops.gemm(GEMM_TN, [(A, Bm, Cc)], flags=fl)
                else:
                    ops.gemm(GEMM_TN, [(A[k0:k1], Bm[k0:k1], slabs[i]) for i, (k0, k1) in enumerate(bounds)], flags=fl)
                    ops.slab_reduce(slabs, Cc)
            row['ns=%d %s' % (len(bounds), tname)] = round(timeit(run), 1)
    best = min(row, key=row.get)
    gf = 2.0 * M * N * K / 1e9
    out['TN M=%d N=%d K=%d' % (M, N, K)] = {'best': best, 'best_us': row[best], 'TFLOPs': round(gf / row[best] * 1e3, 1), 'ns=1 default': row['ns=1 default'],
                                            'all': row}
print(json.dumps(out, indent=1))
