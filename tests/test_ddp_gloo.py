"""CPU, world_size 2 over gloo: the data-parallel path of dlsg_amd.Trainer (bucketed gradient all-reduce + 1/world
folded into Adam) equals the single-process mean of the two shards' gradients (DDP mean-of-means, run_gun.py:63-64).
Kernels are the test-only emulation; what is under test is the sharding / bucketing / reduction logic."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build():
    import dlsg_amd
    from emul_ops import EmulOps
    from helpers import load_case, weights_and_inputs
    args, vocab, g, kind = load_case('small_msrvtt')       # batch 4 -> two shards of 2
    torch.manual_seed(0)
    net = dlsg_amd.CapGnnModel(args, vocab).eval()
    net.set_ops(EmulOps())
    sd, frames, regions, caps, lens = weights_and_inputs(net, g, args)
    net.load_state_dict(sd)
    return net, frames, regions, caps, lens


def _worker(rank, world, port, out_dir):
    for p in (HERE, os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), 'd-lsg-video-caption_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import dlsg_amd
    net, frames, regions, caps, lens = _build()
    sl = slice(rank * 2, rank * 2 + 2)
    tr = dlsg_amd.Trainer(net, world_size=world)
    loss = tr.step(frames[sl], regions[sl], caps[sl], lens[sl], 1.0)
    np.save(os.path.join(out_dir, 'flat%d.npy' % rank), net._flat.numpy())
    np.save(os.path.join(out_dir, 'loss%d.npy' % rank), loss.numpy())
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_step_equals_mean_of_shard_gradients(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    f0, f1 = np.load(tmp_path / 'flat0.npy'), np.load(tmp_path / 'flat1.npy')
    assert np.array_equal(f0, f1)                           # replicas stay bit-identical
    # single process: gradients of each shard, averaged, one Adam step
    import dlsg_amd
    net, frames, regions, caps, lens = _build()
    tr = dlsg_amd.Trainer(net, lr=0.0)
    grads = []
    for r in range(2):
        sl = slice(r * 2, r * 2 + 2)
        tr.step(frames[sl], regions[sl], caps[sl], lens[sl], 1.0)
        grads.append(net._gflat.clone())
    net2, *_ = _build()
    tr2 = dlsg_amd.Trainer(net2)
    net2._gflat.copy_((grads[0] + grads[1]))
    tr2.t = 1
    net2.ops.adam(net2._flat, net2._gflat, tr2.m, tr2.v, tr2.lr, 0.5, 0.9, 1e-8, 1, 0.5)
    assert np.abs(net2._flat.numpy() - f0).max() <= 1e-6


SHARDS3 = [slice(0, 2), slice(2, 3), slice(3, 4)]        # an uneven last batch: 2 + 1 + 1 clips


def _worker3(rank, world, port, out_dir):
    for p in (HERE, os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), 'd-lsg-video-caption_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import dlsg_amd
    net, frames, regions, caps, lens = _build()
    sl = SHARDS3[rank]
    tr = dlsg_amd.Trainer(net, world_size=world)
    for _ in range(2):                       # two steps: the second one starts from all-reduced weights
        loss = tr.step(frames[sl], regions[sl], caps[sl], lens[sl], 1.0)
    tr.check()                               # collective: every rank agrees that no rank reported a time-out
    np.save(os.path.join(out_dir, 'flat3_%d.npy' % rank), net._flat.numpy())
    dist.destroy_process_group()


@pytest.mark.timeout(400)
def test_three_ranks_with_an_uneven_last_batch(tmp_path):
    """three ranks holding 2 + 1 + 1 clips (the tail of an epoch whose size the world does not divide, utils/data.py:122-130 pads
    by wrapping; here the shards simply differ): every rank's loss is the mean over ITS rows and the gradients are averaged with
    equal weights (DDP's mean of means, run_gun.py:63-64) -- replicas bit-identical after two steps and equal to the single-process
    emulation of that rule"""
    port = _free_port()
    mp.spawn(_worker3, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    f = [np.load(tmp_path / ('flat3_%d.npy' % r)) for r in range(3)]
    assert np.array_equal(f[0], f[1]) and np.array_equal(f[0], f[2])
    import dlsg_amd
    net2, frames, regions, caps, lens = _build()
    tr2 = dlsg_amd.Trainer(net2)
    for step in (1, 2):
        # gradients of each shard at the current weights, then one Adam step on their plain mean
        probe, *_ = _build()
        probe.load_state_dict(net2.state_dict())
        trp = dlsg_amd.Trainer(probe, lr=0.0)
        tot = None
        for sl in SHARDS3:
            trp.step(frames[sl], regions[sl], caps[sl], lens[sl], 1.0)
            tot = probe._gflat.clone() if tot is None else tot + probe._gflat
        net2._gflat.copy_(tot)
        net2.ops.adam(net2._flat, net2._gflat, tr2.m, tr2.v, tr2.lr, 0.5, 0.9, 1e-8, step, 1.0 / 3.0)
    assert np.abs(net2._flat.numpy() - f[0]).max() <= 1e-5      # (two Adam steps on sums taken in another order)


def _worker_torch_ddp(rank, world, port, out_dir):
    for p in (HERE, os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), 'd-lsg-video-caption_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import torch_ref as R
    from torch.nn.parallel import DistributedDataParallel
    net, frames, regions, caps, lens = _build()
    ddp = DistributedDataParallel(net, find_unused_parameters=True)      # run_gun.py:63-64
    opt = torch.optim.Adam(ddp.parameters(), lr=1.6e-4, betas=(0.5, 0.9))
    sl = slice(rank * 2, rank * 2 + 2)
    opt.zero_grad()
    outs = ddp(frames[sl], regions[sl], caps[sl], 26, 1.0)[0]             # run_gun.py:183
    loss = R.ragged_ce(outs, caps[sl], lens[sl])                           # run_gun.py:189-198
    loss.backward()
    opt.step()
    np.save(os.path.join(out_dir, 'ddp_flat%d.npy' % rank), net._flat.detach().numpy())
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_reference_style_loop_with_torch_ddp_wrapper_and_adam(tmp_path):
    """The reference's own loop shape (run_gun.py:63-64,183-198,233-234): DistributedDataParallel(model) +
    loss.backward() + torch.optim.Adam, with the model's autograd bridge underneath -- must equal the Trainer path."""
    port = _free_port()
    mp.spawn(_worker_torch_ddp, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    f0, f1 = np.load(tmp_path / 'ddp_flat0.npy'), np.load(tmp_path / 'ddp_flat1.npy')
    assert np.array_equal(f0, f1)
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    t0 = np.load(tmp_path / 'flat0.npy')
    assert np.abs(t0 - f0).max() <= 2e-6


def _gan_build():
    import dlsg_amd
    from emul_ops import EmulOps
    from helpers import load_gan_case

    def mk(args, vocab):
        m = dlsg_amd.CapGnnModel(args, vocab)
        m.set_ops(EmulOps())
        return m
    args, vocab, g, G, D, frames, regions, caps, lens = load_gan_case('gan_msrvtt', mk, lambda a, v: dlsg_amd.DiscV2(a, v).set_ops(EmulOps()))     # batch 4
    return G, D.eval(), frames, regions, caps, lens, torch.from_numpy(g['eps_gp'])


def _worker_gan(rank, world, port, out_dir):
    for p in (HERE, os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), 'd-lsg-video-caption_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import random
    import dlsg_amd
    G, D, frames, regions, caps, lens, eps = _gan_build()
    sl = slice(rank * 2, rank * 2 + 2)
    it = dlsg_amd.GanTrainer(G, D, num_D=1, world_size=world)
    it.eps_source = lambda k: eps[k][sl]
    random.seed(5)
    outs = [it.iteration(frames[sl], regions[sl], caps[sl], lens[sl], 1.0, 0, i + 1) for i in range(2)]
    np.save(os.path.join(out_dir, 'gflat%d.npy' % rank), G._flat.detach().numpy())
    np.save(os.path.join(out_dir, 'dflat%d.npy' % rank), torch.cat([p.detach().reshape(-1) for p in D.parameters()]).numpy())
    np.save(os.path.join(out_dir, 'lossd%d.npy' % rank), np.array([o['loss_D'] for o in outs]))
    h = it.lambda_handler
    np.save(os.path.join(out_dir, 'lambda%d.npy' % rank), np.array(list(h.cap_list) + [h.state, h.current_lambda, h.current_schedule_step]
                                                                   + [o['cap_loss_record'] for o in outs] + [o['loss_G_record'] for o in outs]))
    np.save(os.path.join(out_dir, 'caploss%d.npy' % rank), np.array([o['cap_loss'] for o in outs]))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_gan_iteration_keeps_generator_and_critic_replicas_identical(tmp_path):
    """GanTrainer with two ranks (run_gun.py wraps both models in DDP, :63-68): critic gradients and generator gradients are
    averaged over the ranks, so both replicas of both models stay bit-identical after two iterations; the recorded critic
    loss is the mean over ranks of the single-process losses of the shards, and GANLambdaHandler is fed the all-reduced
    caption loss (run_gun.py:202-203,212), so its state (`cap_list`, schedule position) is identical on both ranks."""
    port = _free_port()
    mp.spawn(_worker_gan, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert np.array_equal(np.load(tmp_path / 'gflat0.npy'), np.load(tmp_path / 'gflat1.npy'))
    assert np.array_equal(np.load(tmp_path / 'dflat0.npy'), np.load(tmp_path / 'dflat1.npy'))
    # single process: the critic loss of each shard
    import random
    import dlsg_amd
    from dlsg_amd import gan
    G, D, frames, regions, caps, lens, eps = _gan_build()
    random.seed(5)
    with torch.no_grad():
        f_caption, obj, mot, alpha = G(frames, regions, caps, 26, 1.0)          # rows are independent: one call, then slices
    losses = []
    eng = D.engine
    for r in range(2):
        sl = slice(r * 2, r * 2 + 2)
        smask = (caps[sl] > 0).float()
        ws = eng.prepare(caps.device, 2, caps.shape[1], f_caption.shape[2], smask, 4)
        eng.proposals(ws, obj[sl].contiguous(), mot[sl].contiguous(), alpha[sl], smask)
        losses.append(float(eng.update_gradients(ws, caps[sl], f_caption[sl].transpose(0, 1).contiguous(), eps[0][sl].reshape(2), 0)[0]))
    want = 0.5 * (losses[0] + losses[1])
    for r in range(2):
        assert abs(np.load(tmp_path / ('lossd%d.npy' % r))[0] - want) <= 1e-5 * max(1.0, abs(want))
    # the lambda handler saw the same (rank-mean) caption losses on both ranks: same cap_list, same schedule state
    l0, l1 = np.load(tmp_path / 'lambda0.npy'), np.load(tmp_path / 'lambda1.npy')
    assert np.array_equal(l0, l1)
    c0, c1 = np.load(tmp_path / 'caploss0.npy'), np.load(tmp_path / 'caploss1.npy')
    assert not np.array_equal(c0, c1)                       # the shards' own losses do differ ...
    assert np.abs(l0[:2] - 0.5 * (c0 + c1)).max() <= 1e-6   # ... and cap_list holds their mean


def _worker_eval(rank, world, port, out_dir):
    for p in (HERE, os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), 'd-lsg-video-caption_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import json
    import collections
    from dlsg_amd import scoring
    from dlsg_amd.data import distributed_indices
    ids = list(range(100, 111))                                  # 11 test clips: the sampler pads to 12 (one duplicate)
    mine = [ids[i] for i in distributed_indices(len(ids), world, rank, 0, True, 0)]
    part = collections.OrderedDict((vid, 'caption of clip %d' % vid) for vid in mine)
    merged = scoring.merge_rank_results(part)
    json.dump({'mine': mine, 'merged': list(merged.items())}, open(os.path.join(out_dir, 'eval%d.json' % rank), 'w'))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_evaluation_merges_the_ranks_caption_dicts(tmp_path):
    """run_gun.py:268-276: each rank decodes its partition of the test clips, `all_gather_object` + dict merge in rank order
    gives every clip exactly once on every rank (the reference hard-codes world size 4 there; here any size)."""
    import json
    port = _free_port()
    mp.spawn(_worker_eval, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (json.load(open(tmp_path / ('eval%d.json' % r))) for r in range(2))
    assert r0['merged'] == r1['merged']
    assert sorted(k for k, _ in r0['merged']) == list(range(100, 111))
    assert set(r0['mine']) != set(r1['mine']) and set(r0['mine']) | set(r1['mine']) == set(range(100, 111))
    assert all(v == 'caption of clip %d' % k for k, v in r0['merged'])


def _worker_fail(rank, world, port, out_dir):
    for p in (HERE, os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), 'd-lsg-video-caption_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import datetime
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    import time
    from dlsg_amd import comm
    from dlsg_amd.hip import load_library
    t0 = time.time()
    msgs = []
    # (1) rank 0 cannot produce the RCCL id (no device / no librccl in this container): its peers must get the code with the
    #     broadcast and raise too, instead of waiting for an id that never comes
    class NoRccl(object):                      # the library as it behaves where librccl cannot be loaded (DLSG_ENOCOMM = -4)
        def __init__(self, lib):
            self._lib = lib

        def __getattr__(self, name):
            return getattr(self._lib, name)

        def dlsg_comm_unique_id(self, uid):
            return -4
    try:
        comm.RcclComm(world, rank, None, lib=NoRccl(load_library()))
        msgs.append('no error')
    except RuntimeError as e:
        msgs.append(str(e))
    # (2) a hipGraph capture that fails on ONE rank: every rank learns it and takes the fallback together
    import dlsg_amd
    net, frames, regions, caps, lens = _build()
    tr = dlsg_amd.Trainer(net, world_size=world)

    def capture(*a, **k):
        if rank == 1:
            raise RuntimeError('capture refused on rank 1')
        tr._graphs = ['captured on rank 0']
    tr._capture = capture
    err = tr._capture_agreed(frames, regions, caps, lens, False)
    msgs.append('none' if err is None else str(err))
    msgs.append(repr(tr._graphs))
    with open(os.path.join(out_dir, 'fail%d.txt' % rank), 'w') as f:
        f.write('\n'.join(msgs) + '\n%.1f\n' % (time.time() - t0))
    dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_failures_on_one_rank_reach_every_rank(tmp_path):
    """N > 1 failure modes: (1) rank 0 fails to create the RCCL id -> rank 1 raises with rank 0's code instead of blocking in the
    id broadcast; (2) a capture failing on one rank -> `Trainer._capture_agreed` returns an error on every rank and no rank keeps
    graphs (run_gun.py:63-64 / train_debug.py:20: a job in which one rank silently changed its collective schedule hangs)."""
    port = _free_port()
    mp.spawn(_worker_fail, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    out = [open(tmp_path / ('fail%d.txt' % r)).read().split('\n') for r in range(2)]
    for r in range(2):
        assert 'dlsg_comm_unique_id failed on rank 0 with code -4' in out[r][0], out[r]
        assert float(out[r][3]) < 60.0
    assert out[0][1] == 'hipGraph capture failed on another rank' and out[1][1] == 'capture refused on rank 1'
    assert out[0][2] == 'None'


def _worker_guard(rank, world, port, out_dir):
    for p in (HERE, os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), 'd-lsg-video-caption_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import dlsg_amd
    net, frames, regions, caps, lens = _build()
    sl = slice(rank * 2, rank * 2 + 2)
    tr = dlsg_amd.Trainer(net, world_size=world, check_every=0)
    before = net._flat.clone()
    # a time-out on rank 1 only, raised BEHIND the first (decoder) bucket: in the encoder's backward, where the persistent BiLSTM runs
    orig = tr._allreduce
    seen = []

    def allreduce(key):
        seen.append(key)
        orig(key)
        if rank == 1 and len(seen) == 1:
            net.ops._persist_word().fill_(3)
    tr._allreduce = allreduce
    tr.step(frames[sl], regions[sl], caps[sl], lens[sl], 1.0)
    skipped = bool(torch.equal(net._flat, before))
    word = int(net.ops._persist_word().item())
    try:
        tr.check()
        msg = 'no error'
    except RuntimeError as e:
        msg = str(e)
    # the word is cleared by the raise: the next step updates on every rank again
    tr._allreduce = orig
    tr.step(frames[sl], regions[sl], caps[sl], lens[sl], 1.0)
    moved = not torch.equal(net._flat, before)
    with open(os.path.join(out_dir, 'guard%d.txt' % rank), 'w') as f:
        f.write('%s\n%d\n%s\n%s\n%d\n' % (skipped, word, msg, moved, len(seen)))
    np.save(os.path.join(out_dir, 'gflat%d.npy' % rank), net._flat.numpy())
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_a_time_out_on_one_rank_behind_the_first_bucket_skips_adam_on_every_rank(tmp_path):
    """dlsg_adam's guard with several ranks: the persistent kernels' time-out word is max-reduced BEHIND the
    last launch of the backward, so a time-out anywhere in the step -- here on rank 1, after the decoder bucket was handed over --
    makes EVERY rank skip the update; `Trainer.check()` then raises on every rank, and the replicas are still identical after the
    next (clean) step.  (Until round 5 the word rode with the first bucket and this case diverged the replicas.)"""
    port = _free_port()
    mp.spawn(_worker_guard, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    out = [open(tmp_path / ('guard%d.txt' % r)).read().split('\n') for r in range(2)]
    for r in range(2):
        assert out[r][0] == 'True', (r, out[r])                 # no rank updated its weights
        assert out[r][1] == '3', (r, out[r])                    # every rank holds the worst code
        assert 'timed out (code 3)' in out[r][2], (r, out[r])   # every rank raises
        assert out[r][3] == 'True' and int(out[r][4]) >= 3
    assert np.array_equal(np.load(tmp_path / 'gflat0.npy'), np.load(tmp_path / 'gflat1.npy'))
