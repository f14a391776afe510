"""Soak of the GAN iteration (run_gun.py:147-234 on the HIP kernels, every phase a replayed hipGraph): N iterations in one process,
the persistent kernels' time-out words and the losses checked every 100.  Prints one JSON line.
usage: PYTHONFAULTHANDLER=1 python3 tools/gan_soak.py [iterations=1000] [batch=64]"""
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
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
V = 1000
args = dlsg_amd.msvd_shaped(use_visual_gan=True)
torch.manual_seed(0)
random.seed(12)
G = dlsg_amd.CapGnnModel(args, dlsg_amd.make_vocab(V))
G.load_state_dict(synth_state_dict(G.state_dict(), 0))
G = G.cuda().train()
D = dlsg_amd.DiscV2(args, V).cuda().train()
batches = [[t.cuda() for t in synth_batch(args, V, B, 1 + k)] for k in range(4)]
it = dlsg_amd.GanTrainer(G, D, num_D=5, total_step=100)
t0 = time.time()
log = []
for i in range(N):
    frames, regions, caps, lens = batches[i % 4]
    out = it.iteration(frames, regions, caps, lens, dlsg_amd.ss_epsilon(i // 100), i // 100, i % 100 + 1)
    if not all(math.isfinite(out[k]) for k in ('cap_loss', 'loss_G', 'loss_D', 'wasserstein')):
        raise SystemExit('non-finite loss at iteration %d: %s' % (i, out))
    if (i + 1) % 100 == 0:
        G.ops.check_persistent()
        log.append({'iteration': i + 1, 'cap_loss': round(out['cap_loss'], 4), 'loss_D': round(out['loss_D'], 4), 'loss_G': round(out['loss_G'], 4)})
torch.cuda.synchronize()
print(json.dumps({'iterations': N, 'batch': B, 'seconds': round(time.time() - t0, 1), 'ms_per_iteration': round((time.time() - t0) / N * 1e3, 2),
                  'persistent_kernel_timeouts': 0, 'every_100': log}))
