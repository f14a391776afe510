"""GPU (-m gpu): the RCCL code paths of dlsg_amd.Trainer executed for real on the one GPU of the test box.

RCCL refuses two ranks on one device, so the collectives run with world_size 1.  Two forms are under test:
(a) comm='rccl' (the default on a GPU): librccl through the C ABI (dlsg_comm_init / dlsg_allreduce_bucket), enqueued on a side
stream forked by an event INSIDE the capture of the step -- the step stays ONE hipGraph with Adam in it;
(b) comm='torch' (backend 'nccl' == RCCL on ROCm):
what is under test is everything the N > 1 bench does around it -- `dist.all_reduce(async_op=True)` on slices of the
gradient arena issued BETWEEN hipGraph segments captured in thread_local mode, the RCCL stream / event ordering against
the next segment replay and against the `fill(gflat, 0)` of the next step, the NCCL watchdog thread alive during
capture and replay, `work.wait()` before the Adam launch.  With one rank the all-reduce is the identity, so three
steps must be bit-identical to the uncut single-graph trainer (and the eager path with collectives to the eager path
without).  Process-group creation comes before any other GPU call of a fresh child process, as in bench.py."""
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


def _worker(rank, port, out_dir):
    for p in (HERE, os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), 'd-lsg-video-caption_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import random
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    import dlsg_amd
    from helpers import load_case, weights_and_inputs

    def build():
        args, vocab, g, kind = load_case('small_msvd')
        torch.manual_seed(0)
        net = dlsg_amd.CapGnnModel(args, vocab)
        sd, frames, regions, caps, lens = weights_and_inputs(net, g, args)
        net.load_state_dict(sd)
        net = net.cuda().train()
        return net, frames.cuda(), regions.cuda(), caps.cuda(), lens

    res = {}
    for name, graphs, coll, comm in (('graph_uncut', True, False, 'auto'), ('graph_rccl', True, True, 'torch'),
                                     ('graph_rccl_ingraph', True, True, 'rccl'), ('eager', False, False, 'auto'),
                                     ('eager_rccl', False, True, 'torch'), ('eager_rccl_abi', False, True, 'rccl')):
        net, frames, regions, caps, lens = build()
        tr = dlsg_amd.Trainer(net, use_graphs=graphs, device_coins=True, comm=comm)
        tr.force_graph_cuts = coll and comm == 'torch'
        tr.force_collectives = coll
        random.seed(5)
        losses = [float(tr.step(frames, regions, caps, lens, 0.8)) for _ in range(3)]
        torch.cuda.synchronize()
        if graphs and coll and comm == 'torch':
            assert len(tr._graphs) == 6 and not tr._adam_in_graph      # five bucket hand-offs (model.py, _engine_backward) + the tail
        if graphs and coll and comm == 'rccl':
            assert len(tr._graphs) == 1 and tr._adam_in_graph          # collectives captured with the step: one replay
            info = tr.collectives_info()
            assert info['where'] == "inside the step's hipGraph" and info['rccl_version'], info
        if coll and comm == 'rccl':
            assert tr._rccl is not None and tr._rccl.world == 1
            tr.close()
        res[name] = (losses, net._flat.cpu().numpy().copy())
    # the C-ABI communicator on a buffer of the bench's real bucket size, on a side stream, captured into a graph and replayed
    from dlsg_amd.comm import RcclComm
    c = RcclComm(1, 0)
    buf = torch.full((45 * 1024 * 1024,), 3.0, device='cuda')
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    c.allreduce([buf], st)
    c.allreduce([buf[:1024], buf[4096:8192]], st)            # grouped form
    st.synchronize()
    assert float(buf[0]) == 3.0 and float(buf[-1]) == 3.0
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        g.capture_begin(capture_error_mode='thread_local')
        buf.mul_(2.0)
        c.allreduce([buf], st)
        g.capture_end()
    g.replay(); g.replay()
    torch.cuda.synchronize()
    assert float(buf[0]) == 12.0 and float(buf[-1]) == 12.0
    c.close()
    # a collective of the bench's real bucket size, to be sure RCCL moved data through its own kernels
    big = torch.ones(45 * 1024 * 1024, device='cuda')
    w = dist.all_reduce(big, async_op=True)
    w.wait()
    torch.cuda.synchronize()
    assert float(big[0]) == 1.0 and float(big[-1]) == 1.0
    np.savez(os.path.join(out_dir, 'res.npz'), **{k + '_flat': v[1] for k, v in res.items()},
             **{k + '_loss': np.array(v[0]) for k, v in res.items()})
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_rccl_all_reduce_between_graph_segments_world1(tmp_path):
    mp.spawn(_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    r = dict(np.load(tmp_path / 'res.npz'))
    assert np.array_equal(r['graph_rccl_loss'], r['graph_uncut_loss'])
    assert np.array_equal(r['graph_rccl_flat'], r['graph_uncut_flat'])
    assert np.array_equal(r['graph_rccl_ingraph_loss'], r['graph_uncut_loss'])
    assert np.array_equal(r['graph_rccl_ingraph_flat'], r['graph_uncut_flat'])
    assert np.array_equal(r['eager_rccl_loss'], r['eager_loss'])
    assert np.array_equal(r['eager_rccl_flat'], r['eager_flat'])
    assert np.array_equal(r['eager_rccl_abi_flat'], r['eager_flat'])
    assert np.abs(r['graph_rccl_flat'] - r['eager_flat']).max() <= 1e-4
