"""usage: python3 tools/pass_bench.py [clips ...] -- the graph-attention pass of SURVEY.md 8(d) in isolation."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
from dlsg_amd.hip import HipOps  # noqa: E402
from dlsg_amd.passbench import run_graph_attention_pass  # noqa: E402

ops = HipOps()
for B in [int(x) for x in sys.argv[1:]] or [256, 1024, 2048]:
    r = run_graph_attention_pass(ops, B=B)
    print(r, '-> %.1f%% of 8 TB/s' % (r['achieved_GBps'] / 80.0))
