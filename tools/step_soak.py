"""Soak of the plain train step (replayed hipGraphs; each step copies one of four resident batches into the graph's static inputs, scheduled sampling with the reference's epsilon schedule, four alternating
batches): N steps in one process, the loss read back and the persistent kernels' time-out word checked every 100 steps.
Prints one JSON line.  usage: PYTHONFAULTHANDLER=1 python3 tools/step_soak.py [steps=3000] [batch=64]"""
import faulthandler
import json
import math
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
import torch  # noqa: E402
import dlsg_amd  # noqa: E402
from dlsg_amd.synth import synth_state_dict, synth_batch  # noqa: E402

faulthandler.enable()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
V = 1000
args = dlsg_amd.msvd_shaped()
torch.manual_seed(0)
random.seed(12)
net = dlsg_amd.CapGnnModel(args, dlsg_amd.make_vocab(V))
net.load_state_dict(synth_state_dict(net.state_dict(), 0))
net = net.cuda().train()
batches = [[t.cuda() for t in synth_batch(args, V, B, 1 + k)] for k in range(4)]
tr = dlsg_amd.Trainer(net, use_graphs=True)       # (until round 6 this said Trainer(net): eager launches, host-bound at ~13.8 ms)
t0 = time.time()
log = []
for i in range(N):
    loss = tr.step(*batches[i % 4], dlsg_amd.ss_epsilon(i // 300))
    if (i + 1) % 100 == 0:
        v = float(loss)
        if not math.isfinite(v):
            raise SystemExit('non-finite loss at step %d' % i)
        tr.check()
        log.append({'step': i + 1, 'loss': round(v, 4), 'ss_epsilon': round(dlsg_amd.ss_epsilon(i // 300), 3)})
torch.cuda.synchronize()
print(json.dumps({'steps': N, 'batch': B, 'seconds': round(time.time() - t0, 1), 'ms_per_step': round((time.time() - t0) / N * 1e3, 3),
                  'persistent_kernel_timeouts': 0, 'every_100': log}))
