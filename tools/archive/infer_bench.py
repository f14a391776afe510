"""Inference throughput (BASELINE.json configs[4]): greedy and beam-5 decoding, batch 128, MSVD-shaped, one MI355X."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
for prec in ('fp32', 'x3_all'):
    net.gemm_precision = prec
    for k in (1, 5):
        net.update_beam_size(k)
        with torch.no_grad():
            net(frames, regions, None)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 3
            for _ in range(n):
                ids = net(frames, regions, None)[0]
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print('gemm=%s beam=%d batch=%d: %.1f ms / batch, %.0f clips/s, ids %s' % (prec, k, B, dt * 1e3, B / dt, tuple(ids.shape)))

net.gemm_precision = 'fp32'
net.update_beam_size(1)
gg = dlsg_amd.GreedyGraph(net, frames, regions)
gg(frames, regions)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    ids = gg(frames, regions)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print('gemm=fp32 greedy hipGraph replay batch=%d: %.1f ms / batch, %.0f clips/s' % (B, dt * 1e3, B / dt))

net.update_beam_size(5)
bg = dlsg_amd.BeamGraph(net, frames, regions)
bg(frames, regions)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    ids = bg(frames, regions)[0]
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print('gemm=fp32 beam=5 hipGraph replay batch=%d: %.1f ms / batch, %.0f clips/s' % (B, dt * 1e3, B / dt))
