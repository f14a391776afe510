"""GPU (-m gpu): the data-parallel path of dlsg_amd.Trainer with the REAL kernels, two ranks sharing the one GPU of the
test box over gloo (RCCL refuses two ranks on one device; what is under test is everything around the collective:
bucket ranges of the gradient arena, the all-reduce issued between hipGraph segments while later segments replay,
the waits before Adam, the 1/world factor).  Both launch modes (kernel by kernel, segmented hipGraph replay) must leave
the replicas bit-identical and equal to the single-process mean of the two shards' gradients."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build():
    import dlsg_amd
    from helpers import load_case, weights_and_inputs
    args, vocab, g, kind = load_case('small_msrvtt')       # batch 4 -> two shards of 2
    torch.manual_seed(0)
    net = dlsg_amd.CapGnnModel(args, vocab).eval()
    sd, frames, regions, caps, lens = weights_and_inputs(net, g, args)
    net.load_state_dict(sd)
    return net.cuda(), frames.cuda(), regions.cuda(), caps.cuda(), lens


def _worker(rank, world, port, out_dir, use_graphs):
    for p in (HERE, os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), 'd-lsg-video-caption_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import dlsg_amd
    net, frames, regions, caps, lens = _build()
    sl = slice(rank * 2, rank * 2 + 2)
    tr = dlsg_amd.Trainer(net, world_size=world, use_graphs=use_graphs, comm='torch')      # gloo: host-issued collectives
    for _ in range(2):                                      # second step replays the captured segments
        loss = tr.step(frames[sl].contiguous(), regions[sl].contiguous(), caps[sl].contiguous(), lens[sl], 1.0)
    torch.cuda.synchronize()
    ngraphs = len(tr._graphs) if tr._graphs else 0
    np.save(os.path.join(out_dir, 'flat%d.npy' % rank), net._flat.cpu().numpy())
    np.save(os.path.join(out_dir, 'meta%d.npy' % rank), np.array([float(loss), ngraphs]))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize('use_graphs', [False, True])
def test_two_ranks_on_one_gpu_match_mean_of_shard_gradients(tmp_path, use_graphs):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), use_graphs), nprocs=2, join=True)
    f0, f1 = np.load(tmp_path / 'flat0.npy'), np.load(tmp_path / 'flat1.npy')
    assert np.array_equal(f0, f1)                           # replicas stay bit-identical
    if use_graphs:
        # the capture is cut at every bucket hand-off: decoder, the two pre-encoders' bucket, the graph modules without their
        # obj_embed weights, then each of those two weights behind its own product (model.py, _engine_backward) + the closing segment
        assert np.load(tmp_path / 'meta0.npy')[1] == 6
    # single process: two steps, each = Adam on the mean of the two shards' gradients
    import dlsg_amd
    net, frames, regions, caps, lens = _build()
    probe = dlsg_amd.Trainer(net, lr=0.0)
    tr = dlsg_amd.Trainer(net)
    for _ in range(2):
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
