"""What one critic update launches, by ATen operator (torch.profiler over ONE eager update at the bench shape): name, calls, the
input shapes of the most frequent ones.  usage: python3 tools/critic_ops_census.py [batch=64]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
import torch  # noqa: E402
import dlsg_amd  # noqa: E402
from dlsg_amd import gan  # noqa: E402
from dlsg_amd.synth import synth_state_dict, synth_batch  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
V = 1000
args = dlsg_amd.msvd_shaped(use_visual_gan=True)
torch.manual_seed(0)
random.seed(12)
G = dlsg_amd.CapGnnModel(args, dlsg_amd.make_vocab(V))
G.load_state_dict(synth_state_dict(G.state_dict(), 0))
G = G.cuda().train()
D = dlsg_amd.DiscV2(args, V).cuda()
frames, regions, caps, lens = [t.cuda() for t in synth_batch(args, V, B, 1)]
it = dlsg_amd.GanTrainer(G, D, num_D=1, total_step=100, use_graphs=False)
with torch.no_grad():
    f_caption, obj, mot, alpha = G(frames, regions, caps, 26, 1.0)
mask = gan.attention_mask(caps)
for _ in range(2):
    it.train_disc(caps, f_caption, obj, mot, mask, alpha)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    it.train_disc(caps, f_caption, obj, mot, mask, alpha)
    torch.cuda.synchronize()
ev = [e for e in prof.key_averages(group_by_input_shape=True) if e.device_time_total > 0 or e.self_device_time_total > 0]
by = {}
for e in ev:
    if e.self_device_time_total <= 0:
        continue
    d = by.setdefault(e.key, {'calls': 0, 'us': 0.0, 'shapes': {}})
    d['calls'] += e.count
    d['us'] += e.self_device_time_total
    s = str(e.input_shapes)
    d['shapes'][s] = d['shapes'].get(s, 0) + e.count
tot = sum(d['calls'] for d in by.values())
print('ops with device time: %d calls' % tot)
for k, d in sorted(by.items(), key=lambda kv: -kv[1]['calls'])[:40]:
    top = sorted(d['shapes'].items(), key=lambda kv: -kv[1])[:6]
    print('%-38s calls %4d  %8.1f us   %s' % (k[:38], d['calls'], d['us'], '; '.join('%dx %s' % (c, s[:70]) for s, c in top)))
