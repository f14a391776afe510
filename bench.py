#!/usr/bin/env python3
"""Benchmark of the D-LSG hot path on MI355X: clips/sec of one TRAIN step (forward + ragged CE + backward + gradient
all-reduce + Adam) of CapGnnModel on synthetic MSVD-shaped features, as BASELINE.json names it.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): batch 64 clips per GPU, 26 frames x (2048 + 4096) frame features, 16 x 2048
region features, vocab 1000, fp32, dropout active, scheduled-sampling eps = 0.95 (epoch 0).  Weak scaling: the
per-GPU batch is fixed, gradients are summed over ranks by bucketed RCCL all-reduce overlapping the backward.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (fresh child processes,
one per GPU, before anything in the parent touches the GPU) and relays rank 0's line; a world size that does not equal
--gpus, or fewer visible devices than ranks, is an error (non-zero exit), never a silently smaller run.

Rank 0 prints ONE JSON line.  `value` is measured with every matrix product in exact fp32 (`--gemm fp32`, the reference's
arithmetic); the split-bf16 policies are reported beside it under `other_gemm_arithmetic` with their measured gradient
error.  `roofline` is measured live with HIP events around the dominant kernel (the fp32-MFMA GEMM);
`roofline_graph_attention` is the HBM-bound object->frame graph kernel the north_star names; `cpu_baseline` times the
oracle (CPU port of the reference path) on a bounded sample on the host cores; `vs_pytorch_rocm_eager` is BASELINE
configs[1]'s comparator: the same step run by PyTorch-ROCm eager (the oracle's torch modules moved to the GPU) in the
same process; `batch_128` is the N = 1 figure at configs[3]'s per-GPU batch.
"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'd-lsg-video-caption_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak; a bf16x3 product costs 3 MFMA flops per flop
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E spec


def usable_cores():
    """cores this process may really use: affinity mask capped by the cgroup CPU quota (the GPU box reports 256 logical
    CPUs but a container quota far below that; oversubscribing torch's thread pool there is pathologically slow)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, min(n, 32))


def cpu_baseline(seconds_budget=24.0):
    """Oracle (oracle/torch_ref.py, kind 'port') at BASELINE configs[0]: B=8, MSVD-shaped, CPU fp32 -- the train step
    (`value`, all usable cores) and, as BASELINE.md section 3.2 asks, the eval forward and both again on 8 threads.
    Bounded: every column gets a share of the budget; a column's warm-up run is its sample if the host is that slow."""
    import dlsg_amd
    from dlsg_amd.synth import synth_state_dict, synth_batch
    from oracle import torch_ref as R
    cores = usable_cores()
    args = dlsg_amd.msvd_shaped()
    vocab = dlsg_amd.make_vocab(1000)
    torch.manual_seed(0)
    net = R.CapGnnModelRef(args, vocab)
    net.load_state_dict(synth_state_dict(net.state_dict(), 0))
    opt = R.make_optimizer(net)
    B = 8
    frames, regions, caps, lens = synth_batch(args, 1000, B, 1)

    def timed(fn, budget, cap=40):
        t0 = time.time()
        fn()                                       # warm-up (also the sample if the host is slow)
        warm = time.time() - t0
        if warm >= budget / 3:
            return 1, warm
        t0, n = time.time(), 0
        while time.time() - t0 < budget - warm and n < cap:
            fn()
            n += 1
        return n, time.time() - t0

    def train():
        R.train_step(net, opt, frames, regions, caps, lens, 0.95)

    def forward():
        with torch.no_grad():
            net(frames, regions, caps, 26, 1.0)

    cols = {}
    plan = [('train', cores, train, 0.42)]
    plan.append(('forward', cores, forward, 0.12))
    if cores != 8:
        plan += [('train_8_threads', 8, train, 0.34), ('forward_8_threads', 8, forward, 0.12)]
    for name, nthr, fn, share in plan:
        torch.set_num_threads(nthr)
        net.train(fn is train)
        random.seed(12)
        n, dt = timed(fn, seconds_budget * share)
        cols[name] = {'clips_per_s': round(B * n / dt, 3), 'threads': nthr, 'runs': n}
    torch.set_num_threads(cores)
    if cores == 8:
        cols['train_8_threads'], cols['forward_8_threads'] = cols['train'], cols['forward']
    return {'value': cols['train']['clips_per_s'], 'unit': 'clips/s', 'cores': cores, 'kind': 'port',
            'sample': '%d train steps of oracle/torch_ref.py, batch %d, vocab 1000, MSVD-shaped, torch CPU fp32, %d threads'
                      % (cols['train']['runs'], B, cores),
            'sample_detail': 'forward + ragged CE + backward + Adam, 26 x (2048+4096) frames + 16 x 2048 regions per clip',
            'eval_forward': {'value': cols['forward']['clips_per_s'], 'unit': 'clips/s', 'threads': cores,
                             'sample': '%d eval forwards (teacher-forced, dropout off), same batch' % cols['forward']['runs']},
            'threads_8': {'train_clips_per_s': cols['train_8_threads']['clips_per_s'],
                          'eval_forward_clips_per_s': cols['forward_8_threads']['clips_per_s'],
                          'note': 'torch.set_num_threads(8): comparable with the 8-core authoring container (BASELINE.md 3.2)'}}


def gpu_eager_baseline(dev, batch, steps=4, warmup=2):
    """BASELINE configs[1] comparator, a baseline leg like cpu_baseline(): the oracle's torch restatement of the reference
    (oracle/torch_ref.py, kind 'port'; same modules, same Adam) on the GPU, stepped by PyTorch-ROCm's own eager kernels
    (rocBLAS / hipBLASLt / MIOpen, fp32).  Same weights, same batch, train mode, same scheduled-sampling ratio."""
    import dlsg_amd
    from dlsg_amd.synth import synth_state_dict, synth_batch
    from oracle import torch_ref as R
    args = dlsg_amd.msvd_shaped()
    vocab = dlsg_amd.make_vocab(1000)
    torch.manual_seed(0)
    net = R.CapGnnModelRef(args, vocab)
    net.load_state_dict(synth_state_dict(net.state_dict(), 0))
    net = net.to(dev).train()
    opt = R.make_optimizer(net)
    frames, regions, caps, lens = synth_batch(args, 1000, batch, 1)
    frames, regions, caps = frames.to(dev), regions.to(dev), caps.to(dev)
    eps = dlsg_amd.ss_epsilon(0)
    random.seed(12)
    for _ in range(warmup):
        R.train_step(net, opt, frames, regions, caps, lens, eps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        R.train_step(net, opt, frames, regions, caps, lens, eps)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    del net, opt
    torch.cuda.empty_cache()
    return {'ms_per_step': round(ms, 2), 'clips_per_s': round(batch / ms * 1e3, 1), 'steps': steps, 'kind': 'port',
            'what': 'oracle/torch_ref.py modules on cuda:0, torch %s eager (rocBLAS/hipBLASLt), fp32, batch %d, train mode, '
                    'tf eps %.3f' % (torch.__version__, batch, eps)}


def gan_iteration_leg(dev, batch, with_eager=True, iters=10, product=True):
    """SURVEY.md 8(f) rank 1, what train_debug.py really trains: one RunGAN iteration (run_gun.py:147-234 -- no-grad generator
    forward, 5 critic updates with gradient penalty, generator step with the GAN term) at the bench shape.  The generator and the
    DiscV2 critic run on the HIP kernels as hand-written launch schedules replayed from hipGraphs (dlsg_amd/gan.py, critic.py); the comparator is
    the oracle's restatement of the reference (oracle/gan_ref.py, kind 'port') stepped by PyTorch-ROCm eager."""
    import dlsg_amd
    from dlsg_amd.synth import synth_state_dict, synth_batch
    V, num_D = 1000, 5
    args = dlsg_amd.msvd_shaped(use_visual_gan=True)
    vocab = dlsg_amd.make_vocab(V)
    torch.manual_seed(0)
    net = dlsg_amd.CapGnnModel(args, vocab)
    sd = synth_state_dict(net.state_dict(), 0)
    net.load_state_dict(sd)
    critic = dlsg_amd.DiscV2(args, V)
    dsd = {k: v.clone() for k, v in critic.state_dict().items()}
    frames, regions, caps, lens = synth_batch(args, V, batch, 1)
    frames, regions, caps = frames.to(dev), regions.to(dev), caps.to(dev)
    eps = dlsg_amd.ss_epsilon(0)
    out = {}
    if product:
        net, critic = net.to(dev).train(), critic.to(dev).train()
        it = dlsg_amd.GanTrainer(net, critic, num_D=num_D, total_step=100)
        random.seed(12)
        for i in range(3):
            it.iteration(frames, regions, caps, lens, eps, 0, i + 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(iters):
            it.iteration(frames, regions, caps, lens, eps, 0, i + 4)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / iters * 1e3
        launches0 = net.ops.launches
        it.iteration(frames, regions, caps, lens, eps, 0, iters + 4)
        out = {'what': 'RunGAN iteration: no-grad generator forward + %d critic updates (WGAN-GP: forward, input backward, the '
                       'derivative of both along the penalty direction, loss backward -- hand-written schedules, no autograd, no '
                       'vendor GEMM) + generator step with the GAN term; MSVD-shaped, batch %d, fp32' % (num_D, batch),
               'ms_per_iteration': round(ms, 2), 'clips_per_s': round(batch / ms * 1e3, 1), 'iterations': iters,
               'host_issued_kernel_launches_per_iteration': net.ops.launches - launches0}
        del it
    del net, critic
    torch.cuda.empty_cache()
    if with_eager:
        from oracle import torch_ref as R
        from oracle import gan_ref
        eager = R.CapGnnModelRef(args, vocab)
        eager.load_state_dict(sd)
        eager = eager.to(dev).train()
        Dref = gan_ref.DiscV2Ref(args, V)
        Dref.load_state_dict(dsd)
        Dref = Dref.to(dev).train()
        opt_G = R.make_optimizer(eager)
        opt_D = torch.optim.Adam(Dref.parameters(), lr=1.6e-4, betas=(0.5, 0.9))
        eps_gp = [torch.rand(batch, 1, 1, device=dev) for _ in range(num_D)]
        random.seed(12)
        with torch.backends.cudnn.flags(enabled=False):           # train_debug.py:53: the fused RNN has no double backward
            gan_ref.gan_iteration(eager, Dref, opt_G, opt_D, frames, regions, caps, lens, eps, 0.01, num_D, eps_gp)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                gan_ref.gan_iteration(eager, Dref, opt_G, opt_D, frames, regions, caps, lens, eps, 0.01, num_D, eps_gp)
            torch.cuda.synchronize()
        ems = (time.perf_counter() - t0) / 3 * 1e3
        out['vs_pytorch_rocm_eager'] = {'ms_per_iteration': round(ems, 1), 'clips_per_s': round(batch / ems * 1e3, 1), 'kind': 'port',
                                        'what': 'oracle/torch_ref.py + oracle/gan_ref.py modules on cuda:0, torch %s eager' % torch.__version__}
        del eager, Dref, opt_G, opt_D
        torch.cuda.empty_cache()
    return out


def gemm_roofline(prof, nsteps, root=ROOT):
    """`roofline` object of the GEMM kernel symbol with the largest share of the timed step (headline: its dominant launch shape).
    prof: HipOps.prof_summary().
    `traffic` = PMC-measured HBM-side bytes per launch (profiles/traffic.json, keyed by kernel symbol + launch shape), averaged
    over the launches that were timed -- only when the file has EVERY launch shape this kernel ran in the step, else null."""
    gk = max((k for k in prof if k.startswith('gemm_')), key=lambda k: prof[k]['ms_total'], default=None)
    g = prof.get(gk) if gk else None
    if not g or g['ms_total'] <= 0:
        return None
    ach = g['work_total'] / (g['ms_total'] * 1e-3) / 1e12
    is_x3 = 'bf16x3' in gk
    peak = PEAK_BF16_MFMA_TFLOPS / 3.0 if is_x3 else PEAK_FP32_MFMA_TFLOPS
    tj = {}
    try:
        tj = json.load(open(os.path.join(root, 'profiles', 'traffic.json'))).get('per_launch_shape', {})
    except Exception:
        tj = {}
    shapes, tsum, tn, all_known = [], 0.0, 0, True
    for shp, d in sorted(g['shapes'].items(), key=lambda kv: -kv[1]['ms_total']):
        ent = tj.get(gk + ' | ' + shp)
        tb = ent.get('hbm_bytes_per_launch') if ent else None
        if tb is None:
            all_known = False
        else:
            tsum += tb * d['launches']; tn += d['launches']
        a_s = d['work_total'] / (d['ms_total'] * 1e-3) / 1e12
        shapes.append({'shape': shp, 'launches_timed': d['launches'], 'avg_launch_ms': round(d['ms_total'] / d['launches'], 4),
                       'achieved': round(a_s, 2), 'frac': round(a_s / peak, 4), 'traffic': tb,
                       'operand_bytes': d.get('operand_bytes'),
                       'traffic_over_operand_bytes': round(tb / d['operand_bytes'], 2) if (tb and d.get('operand_bytes')) else None})
    # symbol under which rocprofv3 lists this kernel (profiles/*kernel_stats*.csv)
    parts = gk.split('_')              # gemm_{f32|bf16x3}_mfma_{tile}_{nt|nn|tn}
    tile, mode = parts[3], parts[-1]
    tmpl = {'nt': 'false, false', 'nn': 'false, true', 'tn': 'true, true'}.get(mode, '')
    if tile == 'streamk':
        # csrc/gemm_sk.hip: ONE persistent launch of one workgroup per CU, 256 x 256 tiles, the remainder tiles cut between the
        # workgroups inside the launch
        sym = 'gemm_sk_kernel<BM, BN, %s> (BM x BN = 256 x 256, 128 x 256 or 128 x 128)' % tmpl
    elif tile == '128x128':
        sym = '%s_w3<128, 128, %s, 32>' % ('gemm_x3_kernel' if is_x3 else 'gemm_kernel', tmpl)
    elif tile.startswith('256x'):
        # csrc/gemm_big.hip; '256x256+rest': the row panels that come in whole rounds of the CUs on this tile, the rest of the
        # rows on the smaller tiles, both inside the one timed call
        sym = 'gemm_big_kernel<256, %s, %s>' % (tile[4:7], tmpl) + (' + the smaller-tile kernel of the remaining rows' if '+' in tile else '')
    elif tile == '128x64':
        sym = '%s_w3<128, 64, %s, 64>' % ('gemm_x3_kernel' if is_x3 else 'gemm_kernel', tmpl)
    elif tile == '64x64':
        sym = '%s<64, 64, %s, 64>' % ('gemm_x3_kernel' if is_x3 else 'gemm_kernel', tmpl)
    else:
        sym = ('skinny_x3_kernel' if is_x3 else 'skinny_kernel') + ('<false>' if mode == 'nt' else '<true>')
    # the headline figures are those of the symbol's DOMINANT LAUNCH SHAPE (most time per step): a symbol runs launches of very
    # different sizes (the stream-K TN kernel: 1.7-ms deep weight gradients down to 0.3-ms mid-size groups) and "flops per launch
    # / average launch duration" only means something for one shape; the symbol-wide totals (what rocprofv3 --stats averages) are
    # in `symbol_total`, every shape in `launch_shapes`
    top = shapes[0]
    return {'kernel': gk, 'mfma': '3 x v_mfma_f32_32x32x16_bf16 per product; peak = 2500/3' if is_x3 else 'v_mfma_f32_32x32x2_f32',
            'rocprof_symbol': sym,
            'launch_shape': top['shape'],
            'symbol_total': {'achieved': round(ach, 2), 'frac': round(ach / peak, 4), 'launches_timed': g['launches'],
                             'avg_launch_ms': round(g['ms_total'] / g['launches'], 4),
                             'traffic': round(tsum / tn) if (all_known and tn) else None},
            'traffic_note': 'HBM-side bytes/launch from profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the real '
                            'step, keyed by kernel + launch shape; Infinity-Cache hits are counted), mean over the timed launches; null '
                            'unless every launch shape of this kernel is in the file.  operand_bytes counts every group\'s operands on '
                            'their own',
            'bound': 'mfma', 'achieved': top['achieved'], 'peak': round(peak, 1), 'unit': 'TFLOP/s', 'frac': top['frac'],
            'traffic': top['traffic'], 'traffic_over_operand_bytes': top['traffic_over_operand_bytes'],
            'launches_timed': top['launches_timed'], 'avg_launch_ms': top['avg_launch_ms'],
            'ms_per_step_in_this_kernel': round(g['ms_total'] / max(1, nsteps), 3), 'launch_shapes': shapes}


def profile_eager_steps(net, tr, batch, eps, nsteps):
    """HIP events around every heavy launch.  Events cannot be read back from inside a replayed graph, so the same step is
    launched kernel by kernel for a few extra steps (not part of `value`) on the same stream."""
    frames, regions, caps, lens = batch
    use_graphs, tr.use_graphs = tr.use_graphs, False
    tr.step(frames, regions, caps, lens, eps)
    torch.cuda.synchronize()
    net.ops.prof = {}
    net.ops.flop_count = 0.0
    for _ in range(nsteps):
        tr.step(frames, regions, caps, lens, eps)
    torch.cuda.synchronize()
    prof = net.ops.prof_summary()
    prof['__gemm_flops_per_step__'] = net.ops.flop_count / max(1, nsteps)
    net.ops.prof = None
    net.ops.flop_count = None
    tr.use_graphs = use_graphs
    return prof


def launch_mode(tr, requested_graphs):
    """what actually ran: a Trainer built with graph_fallback=True downgrades to eager launches when a capture fails"""
    if not requested_graphs:
        return 'eager (--no-graphs)'
    if tr.use_graphs and tr._graphs is not None:
        return 'hipGraph replay'
    return 'eager (hipGraph capture failed)'


def msrvtt_leg(dev, a):
    """BASELINE configs[2]'s per-GPU step: MSR-VTT-shaped (36 regions, 5 proposals, D = 1536, vocab 10 000), batch 64, the same
    fused train step; with the roofline of ITS dominant kernel."""
    import dlsg_amd
    from dlsg_amd.synth import synth_state_dict, synth_batch
    args, V, B = dlsg_amd.msrvtt_shaped(), 10000, 64
    vocab = dlsg_amd.make_vocab(V)
    torch.manual_seed(0)
    net = dlsg_amd.CapGnnModel(args, vocab)
    net.load_state_dict(synth_state_dict(net.state_dict(), 0))
    net = net.to(dev).train()
    net.gemm_precision = a.gemm
    batch = [t.to(dev) for t in synth_batch(args, V, B, 1)]
    tr = dlsg_amd.Trainer(net, use_graphs=not a.no_graphs, graph_fallback=True)
    eps = dlsg_amd.ss_epsilon(0)
    random.seed(12)
    for _ in range(2):
        tr.step(*batch, eps)
    fb = tr.static_inputs() or batch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = tr.step(*fb, eps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {'what': 'CapGnnModel train step, MSR-VTT-shaped: 26 frames x (2048+4096), 36 x 2048 regions, 5 proposals, D 1536, vocab '
                   '10000, dropout on; BASELINE configs[2] per-GPU shard', 'batch_per_gpu': B, 'n_gpus': 1,
           'ms_per_step': round(1e3 * dt / a.steps, 3), 'clips_per_s': round(B * a.steps / dt, 1), 'steps': a.steps,
           'gemm_arithmetic': a.gemm, 'launch': launch_mode(tr, not a.no_graphs), 'final_loss': round(float(loss), 5)}
    nprof = min(3, a.steps)
    prof = profile_eager_steps(net, tr, fb, eps, nprof)
    out['roofline'] = gemm_roofline(prof, nprof)
    for key, name, label in (('o2v_graph_fwd', 'roofline_graph_attention', 'o2v16_kernel'), ('o2v_graph_bwd', 'roofline_graph_attention_bwd', O2V_BWD_KERNELS)):
        o = prof.get(key)
        if o and o['ms_total'] > 0:
            ach = o['work_total'] / (o['ms_total'] * 1e-3) / 1e9
            out[name] = {'kernel': label, 'bound': 'hbm', 'achieved': round(ach, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                         'frac': round(ach / PEAK_HBM_GBS, 4), 'avg_launch_ms': round(o['ms_total'] / o['launches'], 4)}
    del tr, net, batch, fb
    torch.cuda.empty_cache()
    return out


def inference_leg(dev, a):
    """BASELINE configs[4]: inference at batch 128 on one GPU -- greedy and allennlp-style beam search with beam 5 -- each as
    one replayed hipGraph (GreedyGraph / BeamGraph: encoder + all 26 word steps, no host synchronisation inside)."""
    import dlsg_amd
    from dlsg_amd.synth import synth_state_dict, synth_batch
    args, V, B = dlsg_amd.msvd_shaped(), 1000, 128
    vocab = dlsg_amd.make_vocab(V)
    torch.manual_seed(0)
    net = dlsg_amd.CapGnnModel(args, vocab)
    net.load_state_dict(synth_state_dict(net.state_dict(), 0))
    net = net.to(dev).eval()
    net.gemm_precision = a.gemm
    frames, regions, _, _ = synth_batch(args, V, B, 1)
    frames, regions = frames.to(dev), regions.to(dev)
    out = {'what': 'CapGnnModel inference, MSVD-shaped, batch %d, one hipGraph replay per batch (encoder + 26 word steps); ids of the '
                   'replayed graphs are asserted equal to the kernel-by-kernel path in tests/test_gpu_parity.py' % B,
           'batch': B, 'gemm_arithmetic': a.gemm}
    reps = max(3, min(10, a.steps))
    for name, k, cls in (('greedy', 1, dlsg_amd.GreedyGraph), ('beam5', 5, dlsg_amd.BeamGraph)):
        net.update_beam_size(k)
        n0 = net.ops.launches
        g = cls(net, frames, regions)
        launches = (net.ops.launches - n0) // 2        # the constructor runs the schedule twice: warm-up + capture
        g(g.frames, g.regions)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            res = g(g.frames, g.regions)
        ids = res[0] if isinstance(res, tuple) else res
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        L = net.decoder.max_words
        out[name] = {'beam_size': k, 'ms_per_batch': round(ms, 3), 'clips_per_s': round(B / ms * 1e3, 1), 'replays_timed': reps,
                     'launches_per_batch': launches, 'launches_per_word_step': round(launches / L, 1), 'ids_shape': list(ids.shape),
                     'launch': 'hipGraph replay (%s)' % cls.__name__}
        del g
    del net, frames, regions
    torch.cuda.empty_cache()
    return out


def smi_sclk_mhz():
    """current shader clock from `rocm-smi --showclocks` (a child process: nothing of it touches this process's GPU state); None when
    the tool is missing or says nothing parseable"""
    import re
    import subprocess
    try:
        r = subprocess.run(['rocm-smi', '--showclocks'], capture_output=True, text=True, timeout=10)
        m = re.search(r'sclk[^\n]*?\((\d+)\s*Mhz\)', r.stdout, flags=re.I)
        return int(m.group(1)) if m else None
    except Exception:
        return None


def sustained_leg(dev, a, seconds=10.0, clips=320):
    """The train loop as run_gun.py:147-160 feeds it: EVERY step a new batch, gathered on the device from an HBM-resident feature
    store straight into the replayed graph's static input buffers (dlsg_amd.data.ResidentFeatures.batch(ids, out=...)), captions
    copied from the host, epsilon on the reference's schedule (run_gun.py:136), the loss read back and the persistent kernels'
    word checked every 100 steps -- for >= `seconds` of wall clock.  The capture and the warm-up are outside the clock."""
    import threading
    import numpy as np
    import dlsg_amd
    from dlsg_amd import data as D
    from dlsg_amd.hip import host_to_device
    from dlsg_amd.synth import synth_state_dict
    args, V, B = dlsg_amd.msvd_shaped(), 1000, a.batch
    torch.manual_seed(0)
    net = dlsg_amd.CapGnnModel(args, dlsg_amd.make_vocab(V))
    net.load_state_dict(synth_state_dict(net.state_dict(), 0))
    net = net.to(dev).train()
    net.gemm_precision = a.gemm
    g = torch.Generator(device=dev); g.manual_seed(3)
    T, F, O, R = args.max_frames, args.a_feature_size + args.m_feature_size, args.num_obj, args.region_feature_size
    store = D.ResidentFeatures.from_arrays(torch.randn(clips, T, F, device=dev, generator=g),
                                           torch.randn(clips, T, O, R, device=dev, generator=g), O, dev, ops=net.ops)
    rng = np.random.RandomState(5)
    ncap = clips * 8                                   # 8 captions per clip, lengths 5..26, ids as synth_batch draws them
    lens = rng.randint(5, 27, size=ncap)
    caps = np.zeros((ncap, 26), dtype=np.int64)
    for i, n in enumerate(lens):
        caps[i, :n - 1] = rng.randint(4, V, size=n - 1)
        caps[i, n - 1] = 2
    caps_t = torch.from_numpy(caps)
    vid = rng.randint(0, clips, size=ncap)
    steps_per_epoch = ncap // B
    tr = dlsg_amd.Trainer(net, use_graphs=not a.no_graphs, graph_fallback=True)
    random.seed(12)

    def batch_ids(step):
        ep, k = divmod(step, steps_per_epoch)
        perm = np.random.RandomState(100 + ep).permutation(ncap)[k * B:(k + 1) * B]
        return ep, sorted(perm.tolist(), key=lambda i: vid[i], reverse=True)     # video id descending (utils/data.py:90)
    ep, ids = batch_ids(0)
    f0, r0 = store.batch([int(vid[i]) for i in ids])
    for _ in range(3):                                 # capture + warm
        tr.step(f0, r0, host_to_device(caps_t[ids], torch.int64, dev), lens[ids].tolist(), dlsg_amd.ss_epsilon(0))
    st = tr.static_inputs()
    sf, sr = (st[0], st[1]) if st is not None else (f0, r0)
    clk, stop = [], threading.Event()

    def sample():
        while not stop.wait(2.0):
            clk.append(smi_sclk_mhz())
    th = threading.Thread(target=sample, daemon=True)
    torch.cuda.synchronize()
    th.start()
    t0 = time.perf_counter()
    n, losses, t_stage = 0, [], 0.0
    while True:
        ep, ids = batch_ids(n)
        h0 = time.perf_counter()
        store.batch([int(vid[i]) for i in ids], out=(sf, sr))
        cb = host_to_device(caps_t[ids], torch.int64, dev)
        lb = lens[ids].tolist()
        t_stage += time.perf_counter() - h0
        loss = tr.step(sf, sr, cb, lb, dlsg_amd.ss_epsilon(ep))
        n += 1
        if n % 100 == 0:
            losses.append(round(float(loss), 4))       # (a host synchronisation, as a logging interval has)
            tr.check()
            if time.perf_counter() - t0 >= seconds:
                break
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stop.set()
    # what the per-step staging costs on the device: the two gathers alone, HIP events
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        store.batch([int(vid[i]) for i in ids], out=(sf, sr))
    e1.record()
    torch.cuda.synchronize()
    gather_ms = e0.elapsed_time(e1) / 20
    clk = [c for c in clk if c]
    out = {'what': 'loader-fed loop: every step a new %d-clip batch gathered from a %d-clip HBM-resident store into the graph\'s static '
                   'inputs, eps on the reference schedule, loss read every 100 steps' % (B, clips),
           'seconds': round(dt, 2), 'steps': n, 'ms_per_step': round(1e3 * dt / n, 3), 'clips_per_s': round(B * n / dt, 1),
           'launch': launch_mode(tr, not a.no_graphs), 'device_gather_ms_per_step': round(gather_ms, 4),
           'host_staging_ms_per_step': round(1e3 * t_stage / n, 4),
           'sclk_mhz_samples': clk[:8], 'loss_every_100': losses[:3] + (['...'] if len(losses) > 4 else []) + losses[-1:]}
    del tr, net, store
    torch.cuda.empty_cache()
    return out


def dp_schedule_leg(dev, a, batches=(64,)):
    """BASELINE configs[2] / [3] on ONE GPU: the step as a rank of an N-rank job runs it (`Trainer(rehearse_ranks=8)`: multi-rank
    kernel choices, weight gradients flushed per bucket, every bucket through the RCCL communicator inside the captured step),
    timed with and without a co-tenant of an all-reduce's shape on the side stream (dlsg_comm_rehearsal: 32 x 256 threads
    streaming each bucket for about as long as an 8-rank ring all-reduce over xGMI would take), for three backward policies:
    stream-K on every CU (the default with several ranks: it measured fastest), stream-K on the CUs the collective leaves
    free (dlsg_gemm_args.cu_budget = CUs - 32), tiled kernels only (round 5's unmeasured default)."""
    import dlsg_amd
    from dlsg_amd.synth import synth_state_dict, synth_batch
    args, V = dlsg_amd.msvd_shaped(), 1000
    torch.manual_seed(0)
    net = dlsg_amd.CapGnnModel(args, dlsg_amd.make_vocab(V))
    net.load_state_dict(synth_state_dict(net.state_dict(), 0))
    net = net.to(dev).train()
    net.gemm_precision = a.gemm
    eps = dlsg_amd.ss_epsilon(0)
    cus = net.ops.device_cus()
    out = {'what': 'one rank of an 8-rank job rehearsed on one GPU: in-graph RCCL (world 1), per-bucket flush, BiLSTM backward step by '
                   'step; co-tenant = 32 x 256 threads streaming each bucket on the side stream', 'cus': cus}
    # co-tenant calibration: passes so that a bucket's stand-in lasts what an 8-rank ring all-reduce of it would (2 * 7/8 of the bytes
    # per GPU at ~300 GB/s of xGMI bus bandwidth)
    probe = torch.zeros(45 * 1024 * 1024, device=dev)          # 180 MB: the decoder bucket's size
    net.ops.comm_rehearsal(probe, 32, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    net.ops.comm_rehearsal(probe, 32, 4)
    e1.record()
    torch.cuda.synchronize()
    ms_per_pass = e0.elapsed_time(e1) / 4
    target_ms = 2.0 * 7 / 8 * probe.numel() * 4 / 300e9 * 1e3
    passes = max(1, int(round(target_ms / ms_per_pass)))
    out['cotenant'] = {'workgroups': 32, 'passes': passes, 'alone_ms_per_pass_180MB': round(ms_per_pass, 3),
                       'target_ms_180MB': round(target_ms, 3)}
    del probe
    comm = None                                        # one communicator for all the trainers of this leg
    for B in batches:
        batch = [t.to(dev) for t in synth_batch(args, V, B, 1)]
        res = {}
        for name, sk, budget in (('stream_k_all_cus', True, 0), ('stream_k_cu_budget', True, max(8, cus - 32)), ('tiled_backward', False, 0)):
            for co in (False, True):
                tr = dlsg_amd.Trainer(net, use_graphs=not a.no_graphs, graph_fallback=True, rehearse_ranks=8)
                tr._rccl = comm
                default = (net.stream_k_in_backward, net.sk_backward_cu_budget) == (sk, budget)
                net.stream_k_in_backward, net.sk_backward_cu_budget = sk, budget
                tr.rehearse_cotenant = dict(workgroups=32, passes=passes) if co else None
                random.seed(12)
                for _ in range(2):
                    tr.step(*batch, eps)
                fb = tr.static_inputs() or batch
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    tr.step(*fb, eps)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / a.steps * 1e3
                res[name + ('+cotenant' if co else '')] = round(ms, 3)
                if default:
                    out['multi_rank_default'] = name          # what Trainer(world_size > 1) runs (model.Trainer._use_multi_rank_schedule)
                    out.setdefault('collectives', tr.collectives_info())
                tr.check()
                comm, tr._rccl = tr._rccl, None
                del tr
        out['batch_%d_ms_per_step' % B] = res
        del batch
    if comm is not None:
        torch.cuda.synchronize()
        comm.close()
    del net
    torch.cuda.empty_cache()
    return out


def bucket_timeline(tr, batch, eps, dev, world):
    """Where an N > 1 step spends its exchange: ONE extra kernel-by-kernel step with HIP events at every gradient-bucket hand-off
    (main stream: `ready_us`, since the start of the step), behind its all-reduce on the side stream (`reduced_us`) and around the
    join in front of Adam (`join_wait_us` = what the backward did not hide).  Every rank runs it (it has collectives); the maximum
    over ranks of each figure is reported.  Replayed graphs run the same schedule ~0.85x as long (no launch gaps)."""
    import torch.distributed as dist
    model = tr.model
    marks = []

    def ev(name, stream=None):
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream) if stream is not None else e.record()
        marks.append((name, e))
    orig_allreduce, orig_join = tr._allreduce, tr._join_comm

    def allreduce(key):
        label = key if isinstance(key, str) else ' + '.join(key)
        ev('ready: ' + label)
        orig_allreduce(key)
        if tr._comm_stream is not None and tr._comm_pending:
            ev('reduced: ' + label, tr._comm_stream)

    def join():
        ev('join')
        orig_join()
        ev('joined')
    tr._allreduce, tr._join_comm = allreduce, join
    # the extra pass must not move the weights (the legs behind it would time another model state than the steps before it):
    # no Adam, and the step count is put back
    orig_adam, t_saved = tr._adam, tr.t
    tr._adam = lambda *a_, **k_: None
    try:
        captions = batch[2].contiguous()
        lens = torch.as_tensor(batch[3]).to(device=dev, dtype=torch.int64)
        coins = model._draw_coins(captions.shape[1], False, eps)
        torch.cuda.synchronize()
        ev('start')
        tr._eager_step(batch[0], batch[1], captions, lens, coins, model.next_seed())
        ev('end')
        torch.cuda.synchronize()
    finally:
        tr._allreduce, tr._join_comm = orig_allreduce, orig_join
        tr._adam, tr.t = orig_adam, t_saved
    t0 = marks[0][1]
    rows = [(n, t0.elapsed_time(e) * 1e3) for n, e in marks[1:]]
    names = [n for n, _ in rows]
    vals = torch.tensor([t for _, t in rows], dtype=torch.float64, device=dev)
    dist.all_reduce(vals, op=dist.ReduceOp.MAX)
    at = dict(zip(names, [round(float(x), 1) for x in vals]))
    buckets = []
    for n in names:
        if n.startswith('ready: '):
            k = n[7:]
            buckets.append({'bucket': k, 'ready_us': at[n], 'reduced_us': at.get('reduced: ' + k)})
    return {'what': 'one kernel-by-kernel step, HIP events, max over ranks', 'step_us': at.get('end'), 'buckets': buckets,
            'join_wait_us': round(at['joined'] - at['join'], 1) if 'joined' in at and 'join' in at else None,
            'buckets_MB': tr.collectives_info().get('buckets_MB')}


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (the parent makes no GPU call,
    before or after -- counting devices does not initialise the runtime on this image), one per device, RCCL rendezvous on
    127.0.0.1; rank 0's stdout (the JSON line) is relayed, the exit code is the first failing rank's."""
    import socket
    import subprocess
    ndev = torch.cuda.device_count()
    rehearsal = os.environ.get('DLSG_BENCH_ALL_RANKS_ON_DEVICE0') == '1'      # tests: every rank on device 0 over gloo
    if ndev < n and not rehearsal:
        sys.stderr.write('bench.py: --gpus %d but only %d device(s) visible: refusing to run a smaller job under that label\n' % (n, ndev))
        return 3
    sk = socket.socket()
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
    sk.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0', DLSG_BENCH_SPAWNED='1')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # poll every rank: the first one that fails (device init, out of memory, an assertion) takes the others down -- they would sit
    # in a collective waiting for it -- and its exit code is the job's; an overall time-out does the same
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get('DLSG_BENCH_TIMEOUT_S', '2400'))
    bad = None
    while bad is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            c = p.poll()
            if c is not None and c != 0:
                bad = (r, c)
                break
        if bad is None and time.time() > deadline:
            bad = (-1, 124)
        time.sleep(0.2)
    if bad is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except Exception:
                p.kill()
    reader.join(timeout=10)
    sys.stdout.write(b''.join(c for c in chunks if c).decode())
    sys.stdout.flush()
    rcs = [p.poll() for p in procs]
    if bad is None:
        fails = [(r, c) for r, c in enumerate(rcs) if c != 0]
        bad = fails[0] if fails else None
    if bad is not None:
        sys.stderr.write('bench.py: %s; exit codes %s\n' % ('timed out' if bad[0] < 0 else 'rank %d failed with code %s' % bad, rcs))
        return bad[1] if isinstance(bad[1], int) and bad[1] > 0 else 1
    return 0


_JSON_FD = None


def claim_stdout():
    """The contract is ONE JSON line on stdout.  Libraries write there behind Python's back (RCCL prints a version banner through C
    stdio when a communicator is created -- it would land behind the JSON line when the buffer is flushed at exit): from here on
    file descriptor 1 IS stderr, and the line goes to a private duplicate of the original stdout."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    data = (json.dumps(obj) + '\n').encode()
    if _JSON_FD is None:
        sys.stdout.write(data.decode()); sys.stdout.flush()
        return
    while data:
        n = os.write(_JSON_FD, data)
        data = data[n:]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=64, help='clips per GPU (BASELINE configs[1]: 64)')
    ap.add_argument('--shape', default='msvd', choices=['msvd', 'msrvtt'])
    ap.add_argument('--no-pass', action='store_true', help='skip the isolated graph-attention pass measurement')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--eval-mode', action='store_true', help='dropout off (not the reported configuration)')
    ap.add_argument('--gemm', default='fp32', choices=['fp32', 'x3_bwd', 'x3_all'],
                    help='GEMM arithmetic: exact fp32 MFMA (default, the reference\'s arithmetic), or split-bf16 (3 bf16 '
                         'MFMAs/product) for backward / all products')
    ap.add_argument('--no-eager-baseline', action='store_true', help='skip the PyTorch-ROCm eager comparator leg')
    ap.add_argument('--no-batch128', action='store_true', help='skip the extra N = 1 measurement at 128 clips per GPU')
    ap.add_argument('--no-graphs', action='store_true', help='launch every kernel from Python instead of replaying hipGraphs')
    ap.add_argument('--no-gan', action='store_true', help='skip the GAN-iteration leg (SURVEY.md 8f rank 1)')
    ap.add_argument('--leg', default=None, choices=['gan_eager'],
                    help='(internal) run only this side leg and print its JSON: the parent bench starts it as a child process')
    ap.add_argument('--no-inference', action='store_true', help='skip the inference leg (BASELINE configs[4])')
    ap.add_argument('--no-msrvtt', action='store_true', help='skip the MSR-VTT-shaped batch-64 leg (BASELINE configs[2] per GPU)')
    ap.add_argument('--no-sustained', action='store_true', help='skip the loader-fed >= 10 s leg')
    ap.add_argument('--no-dp-schedule', action='store_true', help='skip the one-GPU rehearsal of a data-parallel rank\'s step')
    ap.add_argument('--comm', default='auto', choices=['auto', 'rccl', 'torch'],
                    help='gradient all-reduce: "rccl" = librccl called through the C ABI and captured inside the step\'s hipGraph; '
                         '"torch" = torch.distributed between graph segments; "auto" = rccl on the nccl backend')
    a = ap.parse_args()
    if a.gpus < 1:
        ap.error('--gpus must be >= 1')

    # dmabuf IPC between the ranks' processes (the host driver supports nothing else): must be in the environment before the
    # first HIP call of this process, i.e. before torch.cuda.set_device below
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if a.leg == 'gan_eager':
        claim_stdout()
        emit(gan_iteration_leg(torch.device('cuda', 0), a.batch, with_eager=True, product=False))
        return
    if 'WORLD_SIZE' not in os.environ and a.gpus > 1:
        sys.exit(spawn_ranks(a.gpus, sys.argv[1:]))
    claim_stdout()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != a.gpus:
        sys.stderr.write('bench.py: WORLD_SIZE = %d but --gpus %d: launch with --nproc-per-node == --gpus (or without a launcher: '
                         'bench.py starts its own ranks)\n' % (world, a.gpus))
        sys.exit(2)
    import dlsg_amd
    from dlsg_amd.synth import synth_state_dict, synth_batch
    # test hooks (tests/test_gpu_bench_two_ranks.py): a 1-GPU box can rehearse the N>1 code path with both ranks on device 0
    # over gloo; the driver's runs use neither variable (one rank per GPU, backend nccl = RCCL)
    rehearsal = os.environ.get('DLSG_BENCH_ALL_RANKS_ON_DEVICE0') == '1'
    if rehearsal:
        local = 0
        # the persistent recurrent kernels need every workgroup of a launch resident at once (one per CU); two processes
        # sharing one device can each hold half of the CUs and wait for the other half: shared-device rehearsals use the
        # per-step schedule
        from dlsg_amd.hip import HipOps
        HipOps.persistent_bilstm = False
    backend = os.environ.get('DLSG_BENCH_BACKEND', 'nccl')
    if not rehearsal and torch.cuda.device_count() <= local:
        sys.stderr.write('bench.py: rank %d has no device %d (%d visible)\n' % (rank, local, torch.cuda.device_count()))
        sys.exit(3)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    pg = None
    comm_info = {'rccl_ranks': 1, 'backend': None}
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        import datetime
        pg_timeout = datetime.timedelta(seconds=float(os.environ.get('DLSG_BENCH_PG_TIMEOUT_S', '600')))
        # a rank that never arrives (or dies inside a collective) must end the job, not hang it: rendezvous and collectives time out
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev, timeout=pg_timeout)
        else:
            dist.init_process_group(backend, timeout=pg_timeout)
        if dist.get_world_size() != a.gpus:
            sys.stderr.write('bench.py: process group has %d ranks, --gpus %d\n' % (dist.get_world_size(), a.gpus))
            sys.exit(2)
        devs = [None] * world
        dist.all_gather_object(devs, '%s:%d' % (socket_hostname(), torch.cuda.current_device()))
        if not rehearsal and len(set(devs)) != world:
            sys.stderr.write('bench.py: ranks share devices %s: one rank per GPU is required\n' % devs)
            sys.exit(2)
        comm_info = {'rccl_ranks': dist.get_world_size(), 'backend': backend, 'devices': devs}
        if backend == 'nccl':
            try:
                comm_info['rccl_version'] = '.'.join(map(str, torch.cuda.nccl.version()))
            except Exception:
                comm_info['rccl_version'] = None

    if a.shape == 'msvd':
        args, V = dlsg_amd.msvd_shaped(), 1000
    else:
        args, V = dlsg_amd.msrvtt_shaped(), 10000
    vocab = dlsg_amd.make_vocab(V)
    torch.manual_seed(0)
    net = dlsg_amd.CapGnnModel(args, vocab)
    net.load_state_dict(synth_state_dict(net.state_dict(), 0))       # identical random-init weights on every rank
    net = net.to(dev)
    net.train(not a.eval_mode)
    net.gemm_precision = a.gemm
    frames, regions, caps, lens = synth_batch(args, V, a.batch, 1 + rank)   # each rank its own shard
    frames, regions, caps, lens = frames.to(dev), regions.to(dev), caps.to(dev), lens.to(dev)
    comm = a.comm
    if comm == 'auto':
        comm = 'rccl' if (world > 1 and backend == 'nccl') else 'torch'
    tr = dlsg_amd.Trainer(net, process_group=pg, world_size=world, use_graphs=not a.no_graphs, graph_fallback=True, comm=comm)
    random.seed(12)                                                  # same coin sequence on all ranks (train_debug.py:34-36)
    eps = dlsg_amd.ss_epsilon(0)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        loss = tr.step(frames, regions, caps, lens, eps)
    # The replayed graphs read the batch from the trainer's static device buffers.  Hand those very buffers to step(), as a
    # loader that gathers each batch straight into them does (dlsg_amd.data.ResidentFeatures.batch(ids, out=...)): the batch is
    # resident in HBM either way; otherwise every step would also pay a 258-MB device-to-device staging copy.
    if tr.static_inputs() is not None:
        frames, regions, caps, lens = tr.static_inputs()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = tr.step(frames, regions, caps, lens, eps)
    barrier()
    dt = time.perf_counter() - t0
    loss_v = float(loss)
    launch = launch_mode(tr, not a.no_graphs)
    comm_info.update(tr.collectives_info())
    # per-kernel HIP-event timing for the roofline objects (every rank runs it: the step has collectives)
    nprof = max(1, min(5, a.steps))
    prof = profile_eager_steps(net, tr, (frames, regions, caps, lens), eps, nprof)
    barrier()
    # the same step under the other GEMM arithmetic policies (informational; `value` is the --gemm policy)
    other = {}
    if world == 1:
        for mode in ('fp32', 'x3_bwd', 'x3_all'):
            if mode == a.gemm:
                continue
            net.gemm_precision = mode
            tr2 = dlsg_amd.Trainer(net, use_graphs=not a.no_graphs, graph_fallback=True)
            tr2.m, tr2.v, tr2.t = tr.m, tr.v, tr.t
            for _ in range(2):
                tr2.step(frames, regions, caps, lens, eps)
            fb = tr2.static_inputs() or (frames, regions, caps, lens)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(a.steps):
                tr2.step(*fb, eps)
            torch.cuda.synchronize()
            other[mode] = {'clips_per_s': round(a.batch * a.steps / (time.perf_counter() - t1), 1)}
            del tr2
        # gradient error of every policy against exact fp32: same weights, batch, dropout seed and coins (lr = 0)
        probe = dlsg_amd.Trainer(net, lr=0.0)
        grads = {}
        for mode in ('fp32', 'x3_bwd', 'x3_all'):
            net.gemm_precision = mode
            net.seed_counter = 1000
            random.seed(99)
            probe.step(frames, regions, caps, lens, eps)
            grads[mode] = net._gflat.clone()
        gs = grads['fp32'].abs().max().item()
        for mode in ('x3_bwd', 'x3_all'):
            tgt = other if mode in other else None
            err = (grads[mode] - grads['fp32']).abs().max().item() / max(gs, 1e-30)
            if tgt is not None:
                tgt[mode]['max_grad_err_rel_to_max_grad_vs_fp32'] = float('%.3g' % err)
        del probe, grads
        net.gemm_precision = a.gemm
    b128 = None
    if world == 1 and not a.no_batch128 and a.batch != 128 and a.shape == 'msvd':
      try:                                   # (a side leg: its failure must not cost the headline line)
        # BASELINE configs[3] runs 128 clips per GPU: its "8 GPUs vs 1" needs the N = 1 number at that batch
        torch.cuda.empty_cache()
        f2, r2, c2, l2 = synth_batch(args, V, 128, 1)
        f2, r2, c2, l2 = f2.to(dev), r2.to(dev), c2.to(dev), l2.to(dev)
        tr3 = dlsg_amd.Trainer(net, use_graphs=not a.no_graphs, graph_fallback=True)
        tr3.m, tr3.v, tr3.t = tr.m, tr.v, tr.t
        for _ in range(2):
            tr3.step(f2, r2, c2, l2, eps)
        if tr3.static_inputs() is not None:
            f2, r2, c2, l2 = tr3.static_inputs()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            tr3.step(f2, r2, c2, l2, eps)
        torch.cuda.synchronize()
        d3 = time.perf_counter() - t1
        b128 = {'clips_per_s': round(128 * a.steps / d3, 1), 'ms_per_step': round(1e3 * d3 / a.steps, 3), 'batch_per_gpu': 128,
                'n_gpus': 1, 'gemm_arithmetic': a.gemm, 'steps': a.steps, 'launch': launch_mode(tr3, not a.no_graphs)}
        del tr3, f2, r2, c2, l2
        torch.cuda.empty_cache()
      except Exception as e:                 # noqa: BLE001
        b128 = {'error': '%s: %s' % (type(e).__name__, e)}
        torch.cuda.synchronize()
    # persistent kernels: no hand-off timed out anywhere above -- checked on EVERY rank before the next collective, and the
    # verdict is shared: one rank raising alone would leave the others in the all-reduce below
    failed = 0
    try:
        tr.check()
    except RuntimeError as e:
        sys.stderr.write('bench.py: rank %d: %s\n' % (rank, e))
        failed = 1
    per_rank_ms = [round(1e3 * dt / a.steps, 3)]
    timeline = None
    if world > 1:
        import torch.distributed as dist
        flag = torch.tensor([failed], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        failed = int(flag.item())
    if failed:
        sys.exit(4)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        every = [None] * world
        dist.all_gather_object(every, per_rank_ms[0])
        per_rank_ms = every
        timeline = bucket_timeline(tr, (frames, regions, caps, lens), eps, dev, world)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        n_clips = a.batch * world * a.steps
        out = {
            'metric': 'clips/sec (train step, 26x(2048+4096) feats)', 'value': round(n_clips / dt, 2), 'unit': 'clips/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(1e3 * dt / a.steps, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32' if a.gemm == 'fp32' else 'f32+bf16x3', 'data': 'synthetic',
            # (the driver's record keeps 120 characters of a string: what discriminates the workload comes first)
            'config': {'workload': 'vocab %d, dropout %s, tf eps %.3f, batch %d/GPU; %s-shaped 26x(2048+4096) + %dx2048 regions; train step%s'
                                   % (V, 'off' if a.eval_mode else 'on', eps, a.batch, a.shape.upper(), args.num_obj,
                                      ' + RCCL all-reduce' if world > 1 else ''),
                       'workload_detail': 'CapGnnModel train step: forward + ragged CrossEntropy + backward + Adam%s, one hipGraph replay per step'
                                          % (' + bucketed RCCL gradient all-reduce inside the graph' if world > 1 else ''),
                       'batch_per_gpu': a.batch, 'global_batch': a.batch * world, 'parallelism': 'dp%d' % world,
                       'launch': launch, 'inputs': 'resident in the replayed graphs\' static device buffers (where the HBM-resident '
                                                   'feature store gathers a batch); no per-step staging copy in the timed region',
                       'gemm_arithmetic': a.gemm, 'final_loss': round(loss_v, 5)},
            'rccl_ranks': comm_info.get('rccl_ranks', 1), 'collectives': comm_info,
        }
        if world > 1:
            out['per_rank_ms_per_step'] = per_rank_ms
            out['bucket_timeline'] = timeline
        gemm_flops = prof.pop('__gemm_flops_per_step__', None)
        out['kernel_time_ms_per_step'] = {k: round(v['ms_total'] / nprof, 3) for k, v in prof.items()}
        rl = gemm_roofline(prof, nprof)
        if rl:
            out['roofline'] = rl
        sk = {}
        for k in sorted(prof):
            if 'streamk' in k:
                r1 = gemm_roofline({k: prof[k]}, nprof)
                st_ = r1['symbol_total']          # (per operand layout: the symbol-wide figures; every shape below)
                sk[k] = {'achieved': st_['achieved'], 'peak': r1['peak'], 'unit': r1['unit'], 'frac': st_['frac'], 'traffic': st_['traffic'],
                         'avg_launch_ms': st_['avg_launch_ms'], 'ms_per_step_in_this_kernel': r1['ms_per_step_in_this_kernel'],
                         'launch_shapes': r1['launch_shapes']}
        if sk:
            out['roofline_stream_k'] = dict(sk, kernel='gemm_sk_kernel<...> (csrc/gemm_sk.hip): each call is ONE persistent launch; one entry per '
                                                        'operand layout, with every launch shape of the step')
        if gemm_flops:
            # what the step EXECUTES: every dlsg_gemm call's 2 M N K as launched (the decoder's K / V projections hoisted out of the word
            # loop, no input gradients for regions / frames) + the MFMA work outside dlsg_gemm: the persistent BiLSTM recurrence
            # (forward and backward through time, 2 B 4H H per step and direction) and the object->frame graph kernels
            Hh = args.visual_hidden_size
            Tt = args.max_frames
            bil = 2.0 * a.batch * 4 * Hh * Hh * Tt * 2 * 2 if bool(getattr(net.ops, 'persistent_bilstm', False)) else 0.0
            graph = a.batch * 2 * (44.3e6 + 133e6) * (args.num_obj / 16.0)
            tot = gemm_flops + bil + graph
            out['whole_step_mfma_frac'] = {'executed_flops_per_step': tot, 'dlsg_gemm_calls': gemm_flops, 'persistent_bilstm': bil,
                                           'graph_kernels': graph, 'TFLOPs': round(tot / (dt / a.steps) / 1e12, 2), 'peak': PEAK_FP32_MFMA_TFLOPS,
                                           'frac': round(tot / (dt / a.steps) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                                           'flops_per_clip': round(tot / a.batch / 1e9, 3)}
        tj = {}
        try:
            tj = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json'))).get('per_launch_shape', {})
        except Exception:
            tj = {}

        def o2v_traffic(key):
            shp = list(prof[key]['shapes'])
            ent = tj.get(key + ' | ' + shp[0]) if len(shp) == 1 else None
            return ent.get('hbm_bytes_per_launch') if ent else None
        o = prof.get('o2v_graph_fwd')
        if o and o['ms_total'] > 0:
            ach = o['work_total'] / (o['ms_total'] * 1e-3) / 1e9
            out['roofline_graph_attention'] = {'kernel': 'o2v16_kernel (both encoder streams in one launch) + o2v_combine_kernel', 'bound': 'hbm',
                                               'achieved': round(ach, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                                               'frac': round(ach / PEAK_HBM_GBS, 4), 'traffic': o2v_traffic('o2v_graph_fwd'),
                                               'avg_launch_ms': round(o['ms_total'] / o['launches'], 4)}
        ob = prof.get('o2v_graph_bwd')
        if ob and ob['ms_total'] > 0:
            ach = ob['work_total'] / (ob['ms_total'] * 1e-3) / 1e9
            out['roofline_graph_attention_bwd'] = {'kernel': O2V_BWD_KERNELS, 'bound': 'hbm',
                                                   'achieved': round(ach, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                                                   'frac': round(ach / PEAK_HBM_GBS, 4), 'traffic': o2v_traffic('o2v_graph_bwd'),
                                                   'avg_launch_ms': round(ob['ms_total'] / ob['launches'], 4)}
        if world == 1 and not a.no_pass:
            # the graph-attention pass (object->frame graph x2, LatentPSL x2, self-attention core, decoder
            # attention over cached K',V' x 26 steps) in isolation, 1024 clips in flight, algorithmic bytes 8.79 MB/clip
            from dlsg_amd.passbench import run_graph_attention_pass
            torch.cuda.empty_cache()
            pr = run_graph_attention_pass(net.ops, B=1024)
            ptraffic = None
            try:
                ptraffic = (json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json'))).get('graph_attention_pass_hbm_part_1024') or {}).get(
                    'hbm_bytes_per_launch')
            except Exception:
                ptraffic = None
            hs = pr['hbm_streaming_part']
            try:                                   # the same part with 2048 clips in flight (launch ramp and tail amortised further)
                torch.cuda.empty_cache()
                pr2 = run_graph_attention_pass(net.ops, B=2048)['hbm_streaming_part']
                more = {'clips': 2048, 'ms': pr2['ms'], 'achieved': pr2['achieved_GBps'], 'unit': 'GB/s',
                        'frac': round(pr2['achieved_GBps'] / PEAK_HBM_GBS, 4)}
            except Exception as e:                 # noqa: BLE001
                more = {'error': '%s: %s' % (type(e).__name__, e)}
                torch.cuda.synchronize()
            torch.cuda.empty_cache()
            out['roofline_graph_attention_pass'] = {
                'kernel': 'o2v_fwd x2 + latent_psl_fwd x2 + sa_core_fwd: the HBM-streaming part of the SURVEY.md 8d forward pass '
                          '(4.96 MB/clip), product kernels, 1024 clips in flight', 'bound': 'hbm',
                'achieved': hs['achieved_GBps'], 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                'frac': round(hs['achieved_GBps'] / PEAK_HBM_GBS, 4), 'traffic': ptraffic, 'clips_per_s': hs['clips_per_s'],
                'ms': hs['ms'], 'bytes_per_clip': hs['bytes_per_clip'], 'parts_ms': pr['parts_ms'], 'parts_GBps': pr['parts_GBps'],
                'decoder_term_cache_resident': pr['decoder_term'], 'with_2048_clips_in_flight': more,
                'whole_pass_incl_decoder_term': {'bytes_per_clip': pr['bytes_per_clip'], 'ms': pr['ms'],
                                                 'GBps': pr['achieved_GBps'], 'clips_per_s': pr['clips_per_s']}}
        if b128 is not None:
            out['batch_128'] = b128
        # the side legs must never cost the headline line: a failure in one of them is reported in its field
        def leg(name, fn):
            try:
                out[name] = fn()
            except Exception as e:                                   # noqa: BLE001
                out[name] = {'error': '%s: %s' % (type(e).__name__, e)}
                torch.cuda.synchronize()

        def eager_leg():
            eb = gpu_eager_baseline(dev, a.batch)
            eb['speedup'] = round(out['value'] / eb['clips_per_s'], 2)
            return eb
        side = world == 1 and a.shape == 'msvd' and a.batch == 64
        if side and not a.no_inference:
            leg('inference', lambda: inference_leg(dev, a))
        if side and not a.no_msrvtt:
            leg('msrvtt_b64', lambda: msrvtt_leg(dev, a))
        if side and not a.no_sustained:
            def sus():
                r = sustained_leg(dev, a)
                r['vs_headline_ms_per_step'] = round(r['ms_per_step'] / out['ms_per_step'], 4)
                return r
            leg('sustained', sus)
        if side and not a.no_dp_schedule and not a.no_graphs:
            def dps():
                r = dp_schedule_leg(dev, a, batches=(64, 128))
                r['one_rank_ms_per_step'] = {'batch_64': out['ms_per_step'], 'batch_128': (out.get('batch_128') or {}).get('ms_per_step')}
                return r
            leg('dp_schedule_world1', dps)
        if world == 1 and not a.no_eager_baseline and a.shape == 'msvd':
            leg('vs_pytorch_rocm_eager', eager_leg)
        if world == 1 and not a.no_gan and a.shape == 'msvd' and a.gemm == 'fp32' and not a.no_graphs:
            # the product's GAN iteration runs here, in this process (every launch one of this repo's kernels); only the COMPARATOR
            # -- the oracle stepped by PyTorch-ROCm eager: autograd's double backward over rocBLAS / MIOpen -- runs in a child
            def gan_leg():
                g = gan_iteration_leg(dev, a.batch, with_eager=False)
                if not a.no_eager_baseline:
                    e = child_leg('gan_eager', ['--batch', str(a.batch)], 900)
                    g['vs_pytorch_rocm_eager'] = e.get('vs_pytorch_rocm_eager', e)
                    if 'ms_per_iteration' in g['vs_pytorch_rocm_eager']:
                        g['vs_pytorch_rocm_eager']['speedup'] = round(g['vs_pytorch_rocm_eager']['ms_per_iteration'] / g['ms_per_iteration'], 2)
                return g
            leg('gan_iteration', gan_leg)
        if world == 1 and not a.no_cpu_baseline:
            leg('cpu_baseline', cpu_baseline)
        out['dtype_note'] = {'fp32': 'all products on fp32-input MFMA (exact fp32)',
                             'x3_bwd': 'forward: exact fp32 MFMA; backward products: fp32 operands split into bf16 hi+lo, 3 bf16 '
                                       'MFMAs per product, fp32 accumulate (rel. error ~1e-5)',
                             'x3_all': 'all products: fp32 operands split into bf16 hi+lo, 3 bf16 MFMAs per product, fp32 accumulate'}[a.gemm]
        out['other_gemm_arithmetic'] = other
        # what the contract names first, then the legs the round's review asked for, then the rest (a reader that keeps only the
        # head or the tail of a long line still sees the headline objects)
        first = ['metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                 'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'sustained', 'dp_schedule_world1', 'roofline_graph_attention',
                 'roofline_graph_attention_bwd', 'roofline_graph_attention_pass', 'batch_128', 'msrvtt_b64', 'inference']
        out = {**{k: out[k] for k in first if k in out}, **{k: v for k, v in out.items() if k not in first}}
        emit(out)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        tr.close()
        dist.destroy_process_group()


def child_leg(name, extra, timeout_s):
    """run `bench.py --leg name` as a child process (started with Popen, never exec'd over this process) and return the JSON
    object it prints; a crash or a time-out becomes an {'error': ...} field of the parent's line"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--leg', name] + extra
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {'error': 'child process timed out after %d s' % timeout_s}
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    if r.returncode != 0 or not lines:
        return {'error': 'child process exited with code %d' % r.returncode, 'stderr_tail': r.stderr[-400:]}
    return json.loads(lines[-1])


def socket_hostname():
    import socket
    return socket.gethostname()


O2V_BWD_KERNELS = 'o2v16_bwd_scores_kernel + o2v16_bwd_apply_kernel + o2v_combine_multi_kernel (both encoder streams per launch)'


if __name__ == '__main__':
    main()
