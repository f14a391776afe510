"""Census of the critic's products at the bench shape: every (mode, M, N, K, batch) one critic update sends to the kernel
interface's GEMM (dlsg_amd.gan._Gemm), counted on the CPU with the emulated kernels, then -- on a GPU -- each distinct shape
timed on dlsg_gemm against torch.matmul (rocBLAS).  usage: python tools/critic_gemm_census.py [batch=64] [time=0|1]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
import dlsg_amd  # noqa: E402
from dlsg_amd import gan  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
do_time = len(sys.argv) > 2 and sys.argv[2] == '1'
V, L = 1000, 26
args = dlsg_amd.msvd_shaped(use_visual_gan=True)
P = args.num_proposals


def census():
    from emul_ops import EmulOps
    ops = EmulOps()
    seen = {}
    inner = gan._product

    def product(ops_, mode, A, Bm, C, alpha=1.0, bias=None):
        K = A.shape[-2] if mode == 2 else A.shape[-1]
        nb = C.shape[0] if C.dim() == 3 else 1
        al = all(t.stride(-2) % 4 == 0 and t.data_ptr() % 16 == 0 for t in (A, Bm)) and K % 4 == 0
        key = (mode, C.shape[-2], C.shape[-1], K, nb, bias is not None, al)
        seen[key] = seen.get(key, 0) + 1
        return inner(ops_, mode, A, Bm, C, alpha, bias)
    gan._product = product
    torch.manual_seed(0)
    D = dlsg_amd.DiscV2(args, V).set_ops(ops)
    caps = torch.randint(1, V, (B, L))
    f_caption = torch.randn(B, L, V)
    obj, mot = torch.randn(B, P, 1024), torch.randn(B, P, 1024)
    alpha = torch.rand(B, L, 2 * P)
    mask = gan.attention_mask(caps)
    loss_D = gan.critic_step_losses(D, caps, f_caption, obj, mot, mask, alpha, torch.rand(B, 1, 1))[0]
    loss_D.backward()
    gan._product = inner
    return seen


seen = census()
rows = []
for (mode, M, N, K, nb, bias, al), cnt in sorted(seen.items(), key=lambda kv: -kv[1] * kv[0][1] * kv[0][2] * kv[0][3] * kv[0][4]):
    rows.append({'mode': 'NT NN TN'.split()[mode], 'M': M, 'N': N, 'K': K, 'batch': nb, 'bias': bias, 'aligned': al, 'count': cnt,
                 'MFLOP': round(2e-6 * M * N * K * nb, 1)})
if do_time and torch.cuda.is_available():
    from dlsg_amd.hip import HipOps
    ops = HipOps()
    for r in rows:
        mode, M, N, K, nb = 'NT NN TN'.split().index(r['mode']), r['M'], r['N'], r['K'], r['batch']
        lead = (nb,) if nb > 1 else ()
        A = torch.randn(*lead, *((K, M) if mode == 2 else (M, K)), device='cuda')
        Bm = torch.randn(*lead, *((N, K) if mode == 0 else (K, N)), device='cuda')
        C = torch.empty(*lead, M, N, device='cuda')

        def ours():
            gan._product(ops, mode, A, Bm, C)
        r['kernel'] = ('narrow%d' % ops.gemm_narrow_kind(mode, M, N, K, nb)) if ops.gemm_narrow_kind(mode, M, N, K, nb) else \
            ('tn_split' if mode == 2 and nb == 1 and gan._tn_chunks(M, N, K) > 1 else 'tiled')

        def blas():
            a = A.transpose(-1, -2) if mode == 2 else A
            b = Bm.transpose(-1, -2) if mode == 0 else Bm
            torch.matmul(a, b, out=C)
        res = {}
        for name, fn in (('dlsg_us', ours), ('rocblas_us', blas)):
            for _ in range(3):
                fn()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(20):
                    fn()
            g.replay()
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(5):
                g.replay()
            torch.cuda.synchronize()
            res[name] = round((time.time() - t0) / 100 * 1e6, 1)
        r.update(res)
tot = {'launches': sum(r['count'] for r in rows)}
if do_time and rows and 'dlsg_us' in rows[0]:
    tot['dlsg_ms'] = round(sum(r['count'] * r['dlsg_us'] for r in rows) / 1e3, 3)
    tot['rocblas_ms'] = round(sum(r['count'] * r['rocblas_us'] for r in rows) / 1e3, 3)
for r in rows:
    print(json.dumps(r))
print(json.dumps(tot))
