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
