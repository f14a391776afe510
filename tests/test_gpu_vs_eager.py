"""GPU (-m gpu): BASELINE.json configs[1] -- "batch 64, 1 x MI355X, HIP graph-attention + LSTM decode vs PyTorch-ROCm
eager".  The eager side is the oracle's torch restatement of the reference (same modules, same Adam) moved to the GPU and
stepped with torch's own ROCm kernels; the HIP side is dlsg_amd.Trainer.  Same weights, same batch, train mode.
The measured numbers are written to gpurun_out/vs_eager.json (copied to profiles/ per round)."""
import json
import os
import random
import time

import pytest
import torch

import dlsg_amd

pytestmark = pytest.mark.gpu


def _time(fn, warmup, steps):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def test_train_step_throughput_vs_pytorch_rocm_eager():
    from oracle import torch_ref as R
    from dlsg_amd.synth import synth_state_dict, synth_batch
    B = 64
    args = dlsg_amd.msvd_shaped()
    vocab = dlsg_amd.make_vocab(1000)
    torch.manual_seed(0)
    net = dlsg_amd.CapGnnModel(args, vocab)
    sd = synth_state_dict(net.state_dict(), 0)
    net.load_state_dict(sd)
    frames, regions, caps, lens = synth_batch(args, 1000, B, 1)
    frames, regions, caps = frames.cuda(), regions.cuda(), caps.cuda()

    eager = R.CapGnnModelRef(args, vocab)
    eager.load_state_dict(sd)
    eager = eager.cuda().train()
    opt = R.make_optimizer(eager)
    random.seed(3)
    ms_eager = _time(lambda: R.train_step(eager, opt, frames, regions, caps, lens, 1.0), 2, 5)
    del eager, opt
    torch.cuda.empty_cache()

    out = {'workload': 'CapGnnModel train step, MSVD-shaped, batch 64, train mode, fp32', 'pytorch_rocm_eager_ms': round(ms_eager, 2),
           'pytorch_rocm_eager_clips_per_s': round(B / ms_eager * 1e3, 1), 'torch': torch.__version__}
    for mode in ('fp32', 'x3_bwd'):
        m = dlsg_amd.CapGnnModel(args, vocab)
        m.load_state_dict(sd)
        m = m.cuda().train()
        m.gemm_precision = mode
        tr = dlsg_amd.Trainer(m, use_graphs=True)
        random.seed(3)
        ms = _time(lambda: tr.step(frames, regions, caps, lens.cuda(), 1.0), 4, 10)
        out['hip_%s_ms' % mode] = round(ms, 2)
        out['hip_%s_clips_per_s' % mode] = round(B / ms * 1e3, 1)
        out['speedup_%s' % mode] = round(ms_eager / ms, 2)
        del tr, m
        torch.cuda.empty_cache()
    print(json.dumps(out))
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, 'vs_eager.json'), 'w') as f:
            json.dump(out, f, indent=1)
    except OSError:
        pass
    assert out['speedup_fp32'] >= 1.5, out


def test_gan_iteration_throughput_vs_pytorch_rocm_eager():
    """One RunGAN iteration (run_gun.py:147-234: no-grad generator forward, 5 critic updates with gradient penalty, generator
    step with the GAN term) at batch 64, MSVD-shaped: the oracle's restatement of the reference stepped by PyTorch-ROCm eager
    (cuDNN/MIOpen RNN off, as train_debug.py:53 does for the double backward) against dlsg_amd.GanTrainer."""
    from oracle import torch_ref as R
    from oracle import gan_ref
    from dlsg_amd.synth import synth_state_dict, synth_batch
    B, V, num_D = 64, 1000, 5
    args = dlsg_amd.msvd_shaped(use_visual_gan=True)
    vocab = dlsg_amd.make_vocab(V)
    torch.manual_seed(0)
    net = dlsg_amd.CapGnnModel(args, vocab)
    sd = synth_state_dict(net.state_dict(), 0)
    frames, regions, caps, lens = synth_batch(args, V, B, 1)
    frames, regions, caps = frames.cuda(), regions.cuda(), caps.cuda()
    critic = dlsg_amd.DiscV2(args, V)
    dsd = {k: v.clone() for k, v in critic.state_dict().items()}

    eager = R.CapGnnModelRef(args, vocab)
    eager.load_state_dict(sd)
    eager = eager.cuda().train()
    Dref = gan_ref.DiscV2Ref(args, V)
    Dref.load_state_dict(dsd)
    Dref = Dref.cuda().train()
    opt_G = R.make_optimizer(eager)
    opt_D = torch.optim.Adam(Dref.parameters(), lr=1.6e-4, betas=(0.5, 0.9))
    random.seed(3)
    eps = [torch.rand(B, 1, 1, device='cuda') for _ in range(num_D)]
    with torch.backends.cudnn.flags(enabled=False):
        ms_eager = _time(lambda: gan_ref.gan_iteration(eager, Dref, opt_G, opt_D, frames, regions, caps, lens, 1.0, 0.01, num_D, eps),
                         1, 3)
    del eager, Dref, opt_G, opt_D
    torch.cuda.empty_cache()

    net.load_state_dict(sd)
    net = net.cuda().train()
    critic = critic.cuda().train()
    it = dlsg_amd.GanTrainer(net, critic, num_D=num_D, total_step=100)
    random.seed(3)
    k = [0]

    def step():
        k[0] += 1
        it.iteration(frames, regions, caps, lens, 1.0, 0, k[0])
    ms = _time(step, 3, 8)
    out = {'workload': 'RunGAN iteration (generator forward, 5 critic updates with gradient penalty, generator step), '
                       'MSVD-shaped, batch 64, fp32', 'pytorch_rocm_eager_ms': round(ms_eager, 1), 'hip_ms': round(ms, 1),
           'pytorch_rocm_eager_clips_per_s': round(B / ms_eager * 1e3, 1), 'hip_clips_per_s': round(B / ms * 1e3, 1),
           'speedup': round(ms_eager / ms, 2)}
    print(json.dumps(out))
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, 'gan_vs_eager.json'), 'w') as f:
            json.dump(out, f, indent=1)
    except OSError:
        pass
    assert out['speedup'] >= 1.5, out
