"""GPU (-m gpu): bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, one process per
rank), rehearsed on the 1-GPU test box with both ranks on device 0 over gloo.  Checks the contract of the printed line."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(900)
def test_bench_two_ranks_prints_one_contract_line():
    env = dict(os.environ, DLSG_BENCH_ALL_RANKS_ON_DEVICE0='1', DLSG_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '2',
           '--batch', '16']
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 3 and d['warmup'] == 2 and d['scaling'] == 'weak'
    assert d['unit'] == 'clips/s' and d['higher_is_better'] is True
    assert abs(d['value'] - 2 * 16 * 3 / (d['ms_per_step'] * 3 / 1e3)) <= 0.02 * d['value']   # whole-job aggregate
    assert d['config']['launch'] == 'hipGraph replay'


@pytest.mark.timeout(900)
def test_bench_starts_its_own_ranks_and_refuses_a_smaller_job():
    """`python bench.py --gpus N` without a launcher: the parent starts N fresh rank processes itself and relays rank 0's one
    line (rehearsed with both ranks on device 0 over gloo); with fewer devices than ranks, or a WORLD_SIZE that is not --gpus,
    it exits non-zero instead of timing a smaller job under the label (round-2 review: `--gpus 8` used to run one process and
    print n_gpus 1 with rc 0)."""
    base = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env = dict(base, DLSG_BENCH_ALL_RANKS_ON_DEVICE0='1', DLSG_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '8']
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['rccl_ranks'] == 2 and d['config']['global_batch'] == 16
    assert d['collectives']['backend'] == 'gloo' and len(d['collectives']['devices']) == 2
    # one device, two ranks asked, no rehearsal switch: refuse
    r = subprocess.run(cmd, env=base, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    import torch
    if torch.cuda.device_count() < 2:
        assert r.returncode != 0 and 'refusing' in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    # a launcher's world size that disagrees with --gpus: refuse
    env = dict(base, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 2 and 'WORLD_SIZE' in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
