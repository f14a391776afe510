"""When does each gradient bucket of the train step become ready, and how much of the backward is left to hide its all-reduce?
One rank (force_collectives: the RCCL code path with world size 1), kernel by kernel, HIP events at every bucket hand-off,
at the start of the backward and at the Adam launch.  usage: python3 tools/bucket_timeline.py [batch=64] [msvd|msrvtt]"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
import torch  # noqa: E402
import dlsg_amd  # noqa: E402
from dlsg_amd.synth import synth_state_dict, synth_batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
shape = sys.argv[2] if len(sys.argv) > 2 else 'msvd'
args = dlsg_amd.msvd_shaped() if shape == 'msvd' else dlsg_amd.msrvtt_shaped()
V = 1000 if shape == 'msvd' else 10000
torch.manual_seed(0)
net = dlsg_amd.CapGnnModel(args, dlsg_amd.make_vocab(V))
net.load_state_dict(synth_state_dict(net.state_dict(), 0))
net = net.cuda().train()
batch = [t.cuda() for t in synth_batch(args, V, B, 1)]
tr = dlsg_amd.Trainer(net, use_graphs=False, comm='rccl')
tr.force_collectives = True
marks = []
orig_allreduce, orig_adam, orig_bwd = tr._allreduce, tr._adam, net._engine_backward


def ev(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((name, e))


def allreduce(key):
    ev('bucket ready: %s' % (key if isinstance(key, str) else ' + '.join(key)))
    orig_allreduce(key)


def adam(*a, **k):
    ev('adam')
    orig_adam(*a, **k)


def bwd(*a, **k):
    ev('backward starts')
    return orig_bwd(*a, **k)


tr._allreduce, tr._adam, net._engine_backward = allreduce, adam, bwd
random.seed(12)
rows = []
for it in range(4):
    marks.clear()
    ev('step starts')
    tr.step(*batch, dlsg_amd.ss_epsilon(0))
    ev('step ends')
    torch.cuda.synchronize()
    t0 = marks[0][1]
    rows = [(n, t0.elapsed_time(e) * 1e3) for n, e in marks]
info = tr.collectives_info()
out = {'batch': B, 'shape': shape, 'launch': 'kernel by kernel (eager), one rank, RCCL calls issued (world 1: no data moves)',
       'timeline_us': [{'event': n, 'at_us': round(t, 1)} for n, t in rows], 'buckets_MB': info['buckets_MB']}
print(json.dumps(out, indent=1))
tr.close()
