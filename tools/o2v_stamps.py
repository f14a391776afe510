"""Diagnostic: per-phase shader-clock stamps of the object->frame forward kernel (csrc/o2v16.hip, STAMP build).
    DLSG_O2V_STAMPS=1 python tools/o2v_stamps.py [B]
Prints, per wave of workgroup (0,0), the mean cycles between phase boundaries over the tiles of one launch."""
import math
import os
import sys

os.environ['DLSG_O2V_STAMPS'] = '1'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import ctypes as C  # noqa: E402
from dlsg_amd.hip import HipOps, O2VArgs, _p  # noqa: E402

ops = HipOps()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T, H, O = 26, 1024, 16
NO = T * O
y = torch.tanh(torch.randn(B, NO, H, device='cuda')); v = torch.randn(B, T, H, device='cuda')
g, b_ = torch.ones(H, device='cuda'), torch.zeros(H, device='cuda')
z = torch.empty(B * T, H, device='cuda'); ml = torch.empty(B * T, 2, device='cuda')
st = torch.empty(B * NO, 2, device='cuda'); S = torch.empty(B, NO, T, device='cuda')
ws = torch.zeros(B * (T * H + 64), dtype=torch.float32, device='cuda')
a = O2VArgs()
a.y, a.v, a.g_obj, a.b_obj, a.z, a.ml, a.ostats, a.S = _p(y), _p(v), _p(g), _p(b_), _p(z), _p(ml), _p(st), _p(S)
a.ws, a.ws_bytes = _p(ws), ws.numel() * 4
a.B, a.T, a.NO, a.H, a.nsplit, a.scale, a.eps = B, T, NO, H, 1, 1 / math.sqrt(2048), 1e-5
for _ in range(3):
    rc = ops.lib.dlsg_o2v_fwd(C.byref(a), None)
    assert rc == 0
torch.cuda.synchronize()
tiles = NO // 16
raw = ws.cpu().numpy().view(np.uint64)[:tiles * 16 * 8].reshape(tiles, 16, 8).astype(np.int64)
names = ['issue DMA of the next tile', '-', '-', 'S product + publish', 'barrier (partials)', 'sum partials + softmax',
         'rescale + wait DMA + aggregation || next stats', 'barrier (end of tile)']
d = raw[:, 1:9, :] - raw[:, 0:8, :]
print('B=%d tiles=%d; cycles per phase (mean over tiles 1..%d), per wave 0..7' % (B, tiles, tiles - 2))
for k, n in enumerate(names):
    print('%-30s %s   mean %7.0f' % (n, ' '.join('%6.0f' % x for x in d[1:-1, k, :].mean(0)), d[1:-1, k, :].mean()))
tile = (raw[1:, 0, :] - raw[:-1, 0, :])[1:-1]
print('%-30s %s   mean %7.0f' % ('whole tile', ' '.join('%6.0f' % x for x in tile.mean(0)), tile.mean()))
