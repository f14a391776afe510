import math, os, sys
sys.path.insert(0, 'd-lsg-video-caption_amd')
import torch
from dlsg_amd import hip
lib = os.environ.get('DLSG_LIB')
ops = hip.HipOps()
if lib:
    ops.lib = hip.load_library(lib)
T, H, O, B = 26, 1024, 16, 256
NO = T * O
y = torch.tanh(torch.randn(B, NO, H, device='cuda')); v = torch.randn(B, T, H, device='cuda')
g, b_ = torch.ones(H, device='cuda'), torch.zeros(H, device='cuda')
z = torch.empty(B * T, H, device='cuda'); ml = torch.empty(B * T, 2, device='cuda')
st = torch.empty(B * NO, 2, device='cuda'); S = torch.empty(B, NO, T, device='cuda')
for _ in range(3): ops.o2v_fwd(y, v, g, b_, z, ml, st, S, 1 / math.sqrt(2048), 1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.o2v_fwd(y, v, g, b_, z, ml, st, S, 1 / math.sqrt(2048), 1)
e1.record(); torch.cuda.synchronize()
print(lib, '%.1f us per 256-clip launch -> %.1f us per tile' % (e0.elapsed_time(e1) / 10 * 1e3, e0.elapsed_time(e1) / 10 * 1e3 / 13))
