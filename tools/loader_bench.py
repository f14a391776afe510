"""Loader throughput on the GPU box: a synthetic MSVD-shaped feature set written as HDF5
(`feats` (N,26,6144), `vfeats` (N,26,36,2048)), then
  * ResidentFeatures: one-time upload (host read + H2D) and the per-batch device gather, clips/s
  * StreamedFeatures: worker threads filling a ring of pinned buffers from memory-mapped datasets (and the libhdf5
    hyperslab fallback) + H2D on a side stream, clips/s per worker count (the PCIe-inclusive rate)
usage: python tools/loader_bench.py [N=192] [batch=64]"""
import json
import os
import pickle
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from dlsg_amd import data as D  # noqa: E402
from dlsg_amd.hip import HipOps  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 192
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
d = tempfile.mkdtemp(dir='/tmp')
rng = np.random.RandomState(0)
fp, rp, cp = os.path.join(d, 'f.h5'), os.path.join(d, 'r.h5'), os.path.join(d, 'c.pkl')
t0 = time.time()
D.H5File.create(fp).write('feats', rng.randn(N, 26, 6144).astype(np.float32)).close()
D.H5File.create(rp).write('vfeats', rng.randn(N, 26, 36, 2048).astype(np.float32)).close()
ncap = N * 8
lens = rng.randint(5, 27, size=ncap).tolist()
with open(cp, 'wb') as f:
    pickle.dump(([torch.zeros(26, dtype=torch.long)] * ncap, [torch.zeros(26, dtype=torch.long)] * ncap, lens,
                 rng.randint(0, N, size=ncap).tolist()), f)
out = {'clips': N, 'batch': B, 'write_s': round(time.time() - t0, 2)}
ops = HipOps()
t0 = time.time()
res = D.ResidentFeatures(fp, rp, 16, 'cuda', ops=ops)
torch.cuda.synchronize()
dt = time.time() - t0
out['resident_upload_s'] = round(dt, 2)
out['resident_bytes'] = res.bytes
out['resident_upload_clips_per_s'] = round(N / dt, 1)
ld = D.TrainLoader(cp, res, B, seed=0, drop_last=True)
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.time()
    n = 0
    for batch in ld:
        n += batch[0].shape[0]
    torch.cuda.synchronize()
    dt = time.time() - t0
out['resident_gather_clips_per_s'] = round(n / dt, 1)
out['resident_gather_GBps'] = round(n * (26 * 6144 + 26 * 16 * 2048) * 4 * 2 / dt / 1e9, 1)     # read + write
del res, ld
torch.cuda.empty_cache()
out['host_cpus'] = os.cpu_count()
per_clip = (26 * 6144 + 26 * 16 * 2048) * 4


def streamed(workers, mapped, max_batches):
    st = D.StreamedFeatures(fp, rp, 16, 'cuda', depth=3, workers=workers)
    if not mapped:
        st.mapped = False
    ld = D.TrainLoader(cp, st, B, seed=0, drop_last=True)
    best = 0.0
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.time()
        n = 0
        for i, batch in enumerate(ld):
            n += batch[0].shape[0]
            if i + 1 >= max_batches:
                break
        torch.cuda.synchronize()
        best = max(best, n / (time.time() - t0))
    return best


r = streamed(1, False, 6)
out['streamed_hyperslab_clips_per_s'] = round(r, 1)
out['streamed_mapped'] = {}
for w in (1, 2, 4, 8, 16):
    r = streamed(w, True, 24)
    out['streamed_mapped'][str(w)] = {'clips_per_s': round(r, 1), 'GBps_h2d': round(r * per_clip / 1e9, 2)}
print(json.dumps(out))
