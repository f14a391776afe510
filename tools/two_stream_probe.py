import os, sys, time, random
sys.path.insert(0, 'd-lsg-video-caption_amd')
import torch, dlsg_amd
from dlsg_amd.synth import synth_state_dict, synth_batch
args = dlsg_amd.msvd_shaped(); vocab = dlsg_amd.make_vocab(1000)
torch.manual_seed(0)
for ts in (False, True, False, True):
    net = dlsg_amd.CapGnnModel(args, vocab); net.load_state_dict(synth_state_dict(net.state_dict(), 0)); net = net.cuda().train()
    net.gemm_precision = 'x3_bwd'; net.two_streams = ts
    frames, regions, caps, lens = [x.cuda() for x in synth_batch(args, 1000, 64, 1)]
    tr = dlsg_amd.Trainer(net, use_graphs=True)
    random.seed(12)
    for _ in range(5): tr.step(frames, regions, caps, lens, 0.95)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): tr.step(frames, regions, caps, lens, 0.95)
    torch.cuda.synchronize()
    print('two_streams', ts, (time.perf_counter() - t0) / 20 * 1e3, 'ms/step')
    del tr, net; torch.cuda.empty_cache()
