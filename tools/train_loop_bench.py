"""The whole per-rank training loop on the GPU box, loader included (SURVEY.md 8f rank 2 + the hot path): one epoch of
`Trainer.step` over a synthetic MSVD-shaped feature set written as HDF5, fed by
  * ResidentFeatures (features in HBM, a batch = the device gather into the trainer's static buffers), and
  * StreamedFeatures (memory-mapped host rows -> pinned ring -> H2D on a side stream -> staging copy),
against the same steps on one fixed resident batch (what bench.py times).  Prints one JSON line; clips/s everywhere.
usage: python tools/train_loop_bench.py [N=384] [batch=64] [workers=8]"""
import json
import os
import pickle
import random
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import dlsg_amd  # noqa: E402
from dlsg_amd import data as D  # noqa: E402
from dlsg_amd.synth import synth_state_dict  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 384
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
W = int(sys.argv[3]) if len(sys.argv) > 3 else 8
V = 1000
args = dlsg_amd.msvd_shaped()
d = tempfile.mkdtemp(dir='/tmp')
rng = np.random.RandomState(0)
fp, rp, cp = os.path.join(d, 'f.h5'), os.path.join(d, 'r.h5'), os.path.join(d, 'c.pkl')
D.H5File.create(fp).write('feats', rng.randn(N, 26, 6144).astype(np.float32)).close()
D.H5File.create(rp).write('vfeats', rng.randn(N, 26, 36, 2048).astype(np.float32)).close()
ncap = N * 8
lens = rng.randint(5, 27, size=ncap).tolist()
caps = []
for n in lens:
    c = torch.zeros(26, dtype=torch.long)
    c[:n - 1] = torch.from_numpy(rng.randint(4, V, size=n - 1))
    c[n - 1] = 2
    caps.append(c)
with open(cp, 'wb') as f:
    pickle.dump((caps, [torch.zeros(26, dtype=torch.long)] * ncap, lens, rng.randint(0, N, size=ncap).tolist()), f)

dev = torch.device('cuda', 0)
torch.manual_seed(0)
net = dlsg_amd.CapGnnModel(args, dlsg_amd.make_vocab(V))
net.load_state_dict(synth_state_dict(net.state_dict(), 0))
net = net.to(dev).train()
tr = dlsg_amd.Trainer(net, use_graphs=True)
random.seed(12)
eps = dlsg_amd.ss_epsilon(0)
out = {'clips': N, 'captions': ncap, 'batch': B, 'workers': W, 'host_cpus': os.cpu_count()}


def epoch(loader, into_static):
    torch.cuda.synchronize()
    t0 = time.time()
    n = 0
    if into_static:                                    # the resident store gathers each batch straight into the graph's inputs
        sf, sr, sc, sl = tr.static_inputs()
        caps_set, feats = loader.caps, loader.features
        for b in loader._batches():
            feats.batch([caps_set.video_ids[i] for i in b], out=(sf, sr))
            ib = torch.as_tensor(b, dtype=torch.int64)
            tr.step(sf, sr, caps_set.captions[ib].to(dev, non_blocking=True), [caps_set.lengths[i] for i in b], eps)
            n += len(b)
    else:
        for frames, regions, _, captions, _, cap_lens, _ in loader:
            tr.step(frames, regions, captions, cap_lens, eps)
            n += frames.shape[0]
    torch.cuda.synchronize()
    return n / (time.time() - t0)


res = D.ResidentFeatures(fp, rp, args.num_obj, dev, ops=net.ops)
ld = D.TrainLoader(cp, res, B, seed=0, drop_last=True)
first = next(iter(ld))
for _ in range(3):                                     # capture + warm
    tr.step(first[0], first[1], first[3], first[5], eps)
sf, sr, sc, sl = tr.static_inputs()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(len(ld)):
    tr.step(sf, sr, sc, sl, eps)
torch.cuda.synchronize()
out['fixed_batch_clips_per_s'] = round(len(ld) * B / (time.time() - t0), 1)
out['resident_loader_clips_per_s'] = round(max(epoch(ld, False), epoch(ld, False)), 1)
out['resident_into_static_clips_per_s'] = round(max(epoch(ld, True), epoch(ld, True)), 1)
del res, ld
torch.cuda.empty_cache()
st = D.StreamedFeatures(fp, rp, args.num_obj, dev, depth=3, workers=W)
ld = D.TrainLoader(cp, st, B, seed=0, drop_last=True)
out['streamed_mapped'] = st.mapped
out['streamed_loader_clips_per_s'] = round(max(epoch(ld, False), epoch(ld, False)), 1)
out['streamed_h2d_GBps'] = round(out['streamed_loader_clips_per_s'] * (26 * 6144 + 26 * args.num_obj * 2048) * 4 / 1e9, 2)
print(json.dumps(out))
