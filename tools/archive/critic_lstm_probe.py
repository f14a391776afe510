"""The critic's persistent LSTM launches (csrc/critic_lstm.hip) in both array layouts: time-major (L, n, .) against batch-major
(n, L, .), levels 0 / 1 / 2, for the 192 captions of a critic pass and the 64 of its mixed set.  usage: python3 tools/archive/critic_lstm_probe.py"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
import torch  # noqa: E402
from dlsg_amd import hip  # noqa: E402

ops = hip.HipOps()
L, H = 26, 512
out = {}
for n in (192, 64):
    for bm in (1, 0):
        shp = (n, L) if bm else (L, n)
        t = {k: torch.randn(*shp, w, device='cuda') * 0.1 for k, w in (('addend', 4 * H), ('As', 4 * H), ('Hs', H), ('Cs', H), ('Hprev', H), ('dHs', H),
                                                                    ('dAs', 4 * H), ('dCs', H), ('DA', 4 * H), ('DH', H), ('DC', H), ('gA', 4 * H),
                                                                    ('gC', H), ('gDH', H), ('gDC', H), ('gDHprev', H))}
        W = torch.randn(4 * H, H, device='cuda') / 22
        b1, b2 = torch.zeros(4 * H, device='cuda'), torch.zeros(4 * H, device='cuda')
        nx = int(ops.lib.dlsg_lstm_seq_x_floats(L, n, H))
        xbuf, xbuf2 = torch.empty(nx, device='cuda'), torch.empty(nx, device='cuda')
        flags = torch.empty(int(ops.lib.dlsg_lstm_seq_flag_words(L, n, H)), dtype=torch.int32, device='cuda')
        err = torch.zeros(1, dtype=torch.int32, device='cuda')

        def run(level, hprev=True, bias=True):
            a = hip.LstmSeqArgs()
            for k, v in t.items():
                setattr(a, k, hip._p(v))
            if not hprev:
                a.Hprev, a.gDHprev = None, None
            a.W, a.xbuf, a.xbuf2, a.flags, a.err = hip._p(W), hip._p(xbuf), hip._p(xbuf2), hip._p(flags), hip._p(err)
            a.b_ih, a.b_hh = (hip._p(b1), hip._p(b2)) if bias else (None, None)
            a.L, a.n, a.H, a.batch_major = L, n, H, bm
            rc = ops.lib.dlsg_lstm_seq(C.byref(a), level, ops._stream())
            assert rc == 0, rc
        for level, kw in ((0, {}), (0, dict(hprev=False, bias=False)), (1, {}), (2, {}), (2, dict(hprev=False))):
            for _ in range(3):
                run(level, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run(level, **kw)
            e1.record()
            torch.cuda.synchronize()
            out['n=%d %s level %d %s' % (n, 'batch-major' if bm else 'time-major', level, kw or '')] = round(e0.elapsed_time(e1) / 20 * 1e3, 1)
        assert int(err.item()) == 0
print(json.dumps(out, indent=1))
