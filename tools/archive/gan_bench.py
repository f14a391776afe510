"""One RunGAN iteration (run_gun.py:147-234) at the bench shape: generator and DiscV2 critic on the HIP kernels
(dlsg_amd/gan.py, dlsg_amd/critic.py).  Prints ms per phase (no-grad generator forward, num_D critic updates, generator step
incl. the GAN term) and clips/s of the whole iteration.  usage: python tools/archive/gan_bench.py [batch=64] [iters=8] [num_D=5]"""
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
import torch  # noqa: E402
import dlsg_amd  # noqa: E402
from dlsg_amd import gan  # noqa: E402
from dlsg_amd.synth import synth_state_dict, synth_batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 8
num_D = int(sys.argv[3]) if len(sys.argv) > 3 else 5
V = 1000
args = dlsg_amd.msvd_shaped(use_visual_gan=True)
torch.manual_seed(0)
random.seed(12)
G = dlsg_amd.CapGnnModel(args, dlsg_amd.make_vocab(V))
G.load_state_dict(synth_state_dict(G.state_dict(), 0))
G = G.cuda().train()
D = dlsg_amd.DiscV2(args, V).cuda()
frames, regions, caps, lens = [t.cuda() for t in synth_batch(args, V, B, 1)]
it = dlsg_amd.GanTrainer(G, D, num_D=num_D, total_step=100)
eps = dlsg_amd.ss_epsilon(0)


def sync():
    torch.cuda.synchronize()
    return time.time()


for _ in range(2):
    it.iteration(frames, regions, caps, lens, eps, 0, 1)
t0 = sync()
for i in range(iters):
    it.iteration(frames, regions, caps, lens, eps, 0, i + 1)
whole = (sync() - t0) / iters
# phases, separately timed
smask = (caps > 0).float()
ph = {'generator_forward_nograd': 0.0, 'critic_updates': 0.0, 'generator_step': 0.0}
for i in range(iters):
    t = sync()
    logits_tm, obj, mot, alpha = it.trainer.forward_only(frames, regions, caps, eps, 26, time_major=True)
    t1 = sync()
    it.train_disc(caps, logits_tm, obj, mot, smask, alpha)
    t2 = sync()
    ph['generator_forward_nograd'] += (t1 - t) / iters
    ph['critic_updates'] += (t2 - t1) / iters
ph['generator_step'] = whole - ph['generator_forward_nograd'] - ph['critic_updates']
print(json.dumps({'batch': B, 'num_D': num_D, 'ms_per_iteration': round(whole * 1e3, 2), 'clips_per_s': round(B / whole, 1),
                  'phase_ms': {k: round(v * 1e3, 2) for k, v in ph.items()}}))
