"""Target for the rocprofv3 --pmc passes over the graph-attention pass with 1024 clips in flight:
  cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d <out>/fetch -- python3 tools/pmc_pass_target.py
  cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d <out>/write -- python3 tools/pmc_pass_target.py
The pass runs twice (warm-up + one timed repetition); tools/pmc_pass_sum.py adds the counters of its kernels."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
from dlsg_amd.hip import HipOps  # noqa: E402
from dlsg_amd.passbench import run_graph_attention_pass  # noqa: E402

print(run_graph_attention_pass(HipOps(), B=1024, reps=1))
