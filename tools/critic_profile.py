"""Target for rocprofv3 --kernel-trace --stats: critic updates only (GanTrainer.train_disc) at the bench shape.
usage: python3 tools/critic_profile.py [batch=64] [calls=10] [num_D=5]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
import torch  # noqa: E402
import dlsg_amd  # noqa: E402
from dlsg_amd import gan  # noqa: E402
from dlsg_amd.synth import synth_state_dict, synth_batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 10
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
with torch.no_grad():
    f_caption, obj, mot, alpha = G(frames, regions, caps, 26, 1.0)
logits_tm = f_caption.transpose(0, 1).contiguous()
smask = (caps > 0).float()
import time
for i in range(calls + 2):
    if i == 2:
        torch.cuda.synchronize()
        t0 = time.time()
    it.train_disc(caps, logits_tm, obj, mot, smask, alpha)
torch.cuda.synchronize()
print('ms per critic update: %.2f' % ((time.time() - t0) / calls / num_D * 1e3))
