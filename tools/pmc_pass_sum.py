"""Sum FETCH_SIZE / WRITE_SIZE (KiB per dispatch) of the graph-attention-pass kernels from two rocprofv3 --pmc output dirs.
usage: python3 tools/pmc_pass_sum.py <fetch_dir> <write_dir>"""
import csv
import glob
import json
import sys

KERNELS = ('o2v_partial_kernel', 'o2v_combine_kernel', 'latent_psl_fwd_kernel', 'sa_core_fwd_kernel', 'decatt_fwd_kernel')


def total(d, counter):
    tot, n = 0.0, 0
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter and any(k in r['Kernel_Name'] for k in KERNELS):
                tot += float(r['Counter_Value'])
                n += 1
    return tot, n


fetch, nf = total(sys.argv[1], 'FETCH_SIZE')
write, nw = total(sys.argv[2], 'WRITE_SIZE')
passes = 2
out = {'what': 'graph-attention pass, 1024 clips: 2 x (o2v_partial + o2v_combine), 2 x latent_psl_fwd, sa_core_fwd, 26 x decatt_fwd',
       'dispatches_counted': [nf, nw], 'FETCH_SIZE_KiB': fetch / passes, 'WRITE_SIZE_KiB': write / passes,
       'hbm_bytes_per_launch': int((2 * fetch + write) * 1024 / passes), 'algorithmic_bytes': 8798208 * 1024}
out['ratio'] = round(out['hbm_bytes_per_launch'] / out['algorithmic_bytes'], 2)
print(json.dumps(out))
