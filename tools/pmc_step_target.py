"""Target for the rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over the REAL train step (fp32, kernel by kernel;
batch 64 MSVD-shaped by default: [calls.json] [batch] [msvd|msrvtt]): three eager steps; during the last one every launch that bench.py's roofline objects time is logged in
call order with its launch-shape key, so that tools/pmc_step_traffic.py can attribute the per-dispatch counters.
    cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <out>/fetch -- python3 tools/pmc_step_target.py <out>/calls.json
    cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <out>/write -- python3 tools/pmc_step_target.py
"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'd-lsg-video-caption_amd'))
import torch  # noqa: E402
import dlsg_amd  # noqa: E402
from dlsg_amd.synth import synth_state_dict, synth_batch  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != '-' else None
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
shape_name = sys.argv[3] if len(sys.argv) > 3 else 'msvd'
args = dlsg_amd.msvd_shaped() if shape_name == 'msvd' else dlsg_amd.msrvtt_shaped()
V = 1000 if shape_name == 'msvd' else 10000
torch.manual_seed(0)
net = dlsg_amd.CapGnnModel(args, dlsg_amd.make_vocab(V))
net.load_state_dict(synth_state_dict(net.state_dict(), 0))
net = net.cuda().train()
batch = [t.cuda() for t in synth_batch(args, V, B, 1)]
tr = dlsg_amd.Trainer(net, use_graphs=False)
random.seed(12)
calls = []
for it in range(3):
    if it == 2:
        ops = net.ops
        ops.prof = {}
        ops.prof_min_flops = 0.0          # log EVERY dlsg_gemm call: the counters are attributed by position in the step
        orig = ops._prof_end

        def log(key, e0, work, shape='', *rest, _orig=orig):
            if e0 is not None:
                calls.append({'key': key, 'shape': shape, 'algorithmic_work': work})
            return _orig(key, e0, work, shape, *rest)
        ops._prof_end = log
    tr.step(*batch, dlsg_amd.ss_epsilon(0))
torch.cuda.synchronize()
if out:
    json.dump(calls, open(out, 'w'), indent=1)
print('done', len(calls))
