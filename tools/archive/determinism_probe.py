"""Diagnostic: is one eager train step bit-reproducible?  (same weights, same batch, lr = 0, run N times)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
import dlsg_amd  # noqa: E402
from helpers import load_case, weights_and_inputs  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else 'small_msvd'
args, vocab, g, kind = load_case(tag)
torch.manual_seed(0)
net = dlsg_amd.CapGnnModel(args, vocab).eval()
sd, frames, regions, caps, lens = weights_and_inputs(net, g, args)
net.load_state_dict(sd)
net = net.cuda()
frames, regions, caps = frames.cuda(), regions.cuda(), caps.cuda()
tr = dlsg_amd.Trainer(net, lr=0.0)
ref = None
for i in range(6):
    tr.step(frames, regions, caps, lens, 1.0)
    cur = net._gflat.clone()
    if ref is None:
        ref = cur
    else:
        d = (cur - ref).abs()
        bad = int((d > 0).sum())
        print('run %d: %d differing gradient elements, max %.3g' % (i, bad, d.max().item()))
        if bad:
            G = net.grad_views()
            for k, p in net.named_parameters():
                o = net._offsets[k]
                dd = d[o:o + p.numel()]
                if (dd > 0).any():
                    print('   ', k, int((dd > 0).sum()), dd.max().item())
