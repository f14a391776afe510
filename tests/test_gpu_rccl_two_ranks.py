"""GPU (-m gpu), needs >= 2 devices (self-skips on the 1-GPU test box): the data-parallel train step over RCCL for real --
backend nccl, one rank per device, as the driver's N > 1 bench runs launch it (run_gun.py:63-64, train_debug.py:20).

Asserted for both forms of the exchange (comm='rccl': C-ABI communicator, collectives captured inside the step's hipGraph;
comm='torch': torch.distributed all-reduces between graph segments) and for eager launches:
  * the replicas are bit-identical after three steps;
  * the all-reduced gradient of step 1 equals the single-process mean of the two shards' gradients (DDP mean-of-means) and
    the token-weighted combination equals the gradient of the unsplit batch (as tests/test_gpu_parity.py does on one device);
  * the weights after three steps equal a single-process run that applies Adam to the summed shard gradients with 1/world.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _paths():
    for p in (HERE, os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), 'd-lsg-video-caption_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)


def _build(dev):
    import dlsg_amd
    from helpers import load_case, weights_and_inputs
    args, vocab, g, kind = load_case('small_msrvtt')       # batch 4 -> two shards of 2
    torch.manual_seed(0)
    net = dlsg_amd.CapGnnModel(args, vocab).eval()
    sd, frames, regions, caps, lens = weights_and_inputs(net, g, args)
    net.load_state_dict(sd)
    return net.to(dev), frames.to(dev), regions.to(dev), caps.to(dev), lens


def _worker(rank, world, port, out_dir, use_graphs, comm):
    _paths()
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dev = torch.device('cuda', rank)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)      # before any other GPU call
    import dlsg_amd
    net, frames, regions, caps, lens = _build(dev)
    sl = slice(rank * 2, rank * 2 + 2)
    shard = (frames[sl].contiguous(), regions[sl].contiguous(), caps[sl].contiguous(), lens[sl])
    probe = dlsg_amd.Trainer(net, lr=0.0, world_size=world, use_graphs=False, comm=comm)
    probe.step(*shard, 1.0)                                  # lr 0: the arena is left holding the all-reduced gradient sum
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, 'gsum%d.npy' % rank), net._gflat.cpu().numpy())
    probe.close()
    tr = dlsg_amd.Trainer(net, world_size=world, use_graphs=use_graphs, comm=comm)
    for _ in range(3):
        loss = tr.step(*shard, 1.0)
    torch.cuda.synchronize()
    info = tr.collectives_info()
    np.save(os.path.join(out_dir, 'flat%d.npy' % rank), net._flat.cpu().numpy())
    np.save(os.path.join(out_dir, 'meta%d.npy' % rank), np.array([float(loss), info['graph_replays_per_step']]))
    dist.barrier()
    tr.close()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize('use_graphs,comm', [(True, 'rccl'), (True, 'torch'), (False, 'rccl')])
def test_two_ranks_over_rccl_match_single_process_gradients(tmp_path, use_graphs, comm):
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs: RCCL refuses two ranks on one device (this box has %d)' % torch.cuda.device_count())
    _paths()
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), use_graphs, comm), nprocs=2, join=True)
    f0, f1 = np.load(tmp_path / 'flat0.npy'), np.load(tmp_path / 'flat1.npy')
    assert np.array_equal(f0, f1)                           # replicas stay bit-identical
    g0, g1 = np.load(tmp_path / 'gsum0.npy'), np.load(tmp_path / 'gsum1.npy')
    assert np.array_equal(g0, g1)
    if use_graphs:
        assert np.load(tmp_path / 'meta0.npy')[1] == (1 if comm == 'rccl' else 4)
    # single process on device 0: shard gradients, their sum, the unsplit batch
    import dlsg_amd
    dev = torch.device('cuda', 0)
    net, frames, regions, caps, lens = _build(dev)
    probe = dlsg_amd.Trainer(net, lr=0.0)
    shard_g = []
    for r in range(2):
        sl = slice(r * 2, r * 2 + 2)
        probe.step(frames[sl].contiguous(), regions[sl].contiguous(), caps[sl].contiguous(), lens[sl], 1.0)
        shard_g.append(net._gflat.clone())
    gsum = (shard_g[0] + shard_g[1]).cpu().numpy()
    scale = max(1e-30, float(np.abs(gsum).max()))
    assert np.abs(g0 - gsum).max() <= 1e-6 * scale          # RCCL's sum of two addends == the local sum (one rounding)
    probe.step(frames, regions, caps, lens, 1.0)
    n = [int(lens[r * 2:r * 2 + 2].sum()) for r in range(2)]
    whole = net._gflat.cpu().numpy()
    tok = ((n[0] * shard_g[0] + n[1] * shard_g[1]) / (n[0] + n[1])).cpu().numpy()
    assert np.abs(whole - tok).max() <= 2e-5 * scale
    # three Adam steps on the summed gradients with grad_scale 1/world
    net, frames, regions, caps, lens = _build(dev)
    probe = dlsg_amd.Trainer(net, lr=0.0)
    tr = dlsg_amd.Trainer(net)
    for _ in range(3):
        g = torch.zeros_like(net._gflat)
        for r in range(2):
            sl = slice(r * 2, r * 2 + 2)
            probe.step(frames[sl].contiguous(), regions[sl].contiguous(), caps[sl].contiguous(), lens[sl], 1.0)
            g += net._gflat
        net._gflat.copy_(g)
        tr.t += 1
        net.ops.adam(net._flat, net._gflat, tr.m, tr.v, tr.lr, 0.5, 0.9, 1e-8, tr.t, 0.5)
    torch.cuda.synchronize()
    assert np.abs(net._flat.cpu().numpy() - f0).max() <= 2e-6
