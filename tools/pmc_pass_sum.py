"""Sum FETCH_SIZE / WRITE_SIZE (KiB per dispatch) of the graph-attention-pass kernels from two rocprofv3 --pmc output dirs.
usage: python3 tools/pmc_pass_sum.py <fetch_dir> <write_dir>"""
import csv
import glob
import json
import sys

# the HBM-streaming part of the pass (object->frame graph x2, LatentPSL x2, self-attention core: 4.96 MB/clip);
# the decoder term (dec_mid_fwd x 26) re-reads 134 MB of K', V' that stay in the Infinity Cache and is reported separately
KERNELS = ('o2v16_kernel', 'o2v_partial_kernel', 'o2v_combine_kernel', 'latent_psl_fwd_kernel', 'sa_core_fwd_kernel')
DECODER = ('dec_mid_fwd_kernel',)


def total(d, counter, kernels=KERNELS):
    tot, n = 0.0, 0
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter and any(k in r['Kernel_Name'] for k in kernels):
                tot += float(r['Counter_Value'])
                n += 1
    return tot, n


fetch, nf = total(sys.argv[1], 'FETCH_SIZE')
write, nw = total(sys.argv[2], 'WRITE_SIZE')
passes = 2
out = {'what': 'HBM-streaming part of the graph-attention pass, 1024 clips: o2v16_kernel (both streams in one launch), 2 x latent_psl_fwd, '
               'sa_core_fwd', 'dispatches_counted': [nf, nw], 'FETCH_SIZE_KiB': fetch / passes, 'WRITE_SIZE_KiB': write / passes,
       'hbm_bytes_per_launch': int((2 * fetch + write) * 1024 / passes), 'algorithmic_bytes': 4964352 * 1024}
out['ratio'] = round(out['hbm_bytes_per_launch'] / out['algorithmic_bytes'], 2)
df, ndf = total(sys.argv[1], 'FETCH_SIZE', DECODER)
dw, ndw = total(sys.argv[2], 'WRITE_SIZE', DECODER)
out['decoder_term'] = {'what': "26 x dec_mid_fwd over the K', V' cache of 1024 clips (fabric-side bytes: Infinity-Cache hits included)",
                       'dispatches_counted': [ndf, ndw], 'fabric_bytes': int((2 * df + dw) * 1024 / passes),
                       'algorithmic_bytes': 3833856 * 1024}
print(json.dumps(out))
