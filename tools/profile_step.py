"""Target for rocprofv3 --kernel-trace --stats: N eager train steps (kernel by kernel) in one arithmetic policy.
usage: python3 tools/profile_step.py [x3_bwd|fp32|x3_all] [steps] [batch] [msvd|msrvtt] [eager|graphs]
(graphs: the step as bench.py times it, one hipGraph replay per step -- the kernels' durations without host-side launch gaps)"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
import torch  # noqa: E402
import dlsg_amd  # noqa: E402
from dlsg_amd.synth import synth_state_dict, synth_batch  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'x3_bwd'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
shape = sys.argv[4] if len(sys.argv) > 4 else 'msvd'
graphs = len(sys.argv) > 5 and sys.argv[5] == 'graphs'
args = dlsg_amd.msvd_shaped() if shape == 'msvd' else dlsg_amd.msrvtt_shaped()
V = 1000 if shape == 'msvd' else 10000
vocab = dlsg_amd.make_vocab(V)
torch.manual_seed(0)
net = dlsg_amd.CapGnnModel(args, vocab)
net.load_state_dict(synth_state_dict(net.state_dict(), 0))
net = net.cuda().train()
net.gemm_precision = mode
frames, regions, caps, lens = synth_batch(args, V, B, 1)
frames, regions, caps, lens = frames.cuda(), regions.cuda(), caps.cuda(), lens.cuda()
tr = dlsg_amd.Trainer(net, use_graphs=graphs)
random.seed(12)
for _ in range(steps):
    tr.step(frames, regions, caps, lens, dlsg_amd.ss_epsilon(0))
torch.cuda.synchronize()
print('done', mode, steps)
