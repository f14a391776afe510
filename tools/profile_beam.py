"""Target for rocprofv3 --kernel-trace --stats: inference (BASELINE.json configs[4]), MSVD-shaped, 3 eager calls.
usage: python3 tools/profile_beam.py [batch=128] [beam=5]   (beam=1: greedy)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
import torch  # noqa: E402
import dlsg_amd  # noqa: E402
from dlsg_amd.synth import synth_state_dict, synth_batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
args = dlsg_amd.msvd_shaped()
vocab = dlsg_amd.make_vocab(1000)
torch.manual_seed(0)
net = dlsg_amd.CapGnnModel(args, vocab)
net.load_state_dict(synth_state_dict(net.state_dict(), 0))
net = net.cuda().eval()
frames, regions, caps, lens = synth_batch(args, 1000, B, 1)
frames, regions = frames.cuda(), regions.cuda()
net.update_beam_size(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
with torch.no_grad():
    for _ in range(3):
        ids = net(frames, regions, None)[0]
torch.cuda.synchronize()
print('done', tuple(ids.shape))
