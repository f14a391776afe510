"""Round-3 fixture, generated like the others by running the REFERENCE model (imported from /root/reference) on CPU:

    python tests/golden/make_goldens_r3.py

`full_default_b2`: the reference's DEFAULT feature dimensions -- `--a_feature_size 1536 --m_feature_size 1024`
(utils/opt.py:69-70) with the MSVD overrides of run_gun.py:31-40 -- at full hidden sizes, batch 2.  The other full-size
fixtures use BASELINE.json's 2048 + 4096 frame features; this one pins the 1536-wide strided slice of the 2-D stream
(models/model.py:70) and the 2560-wide frame projection the reference's own defaults produce.  Only arrays are written.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

from make_goldens import run_case                                              # noqa: E402  (imports the reference)
from dlsg_amd.config import make_args, apply_dataset_overrides                 # noqa: E402


def default_dims():
    return apply_dataset_overrides(make_args(dataset='msvd'))                  # A = 1536, M = 1024: the argparse defaults


if __name__ == '__main__':
    args = default_dims()
    assert (args.a_feature_size, args.m_feature_size) == (1536, 1024)
    run_case('full_default_b2', args, V=1000, B=2, seed=23, store_weights=False, store_inter=False, full_logits=False)
