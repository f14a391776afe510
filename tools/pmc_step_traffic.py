"""Attribute the per-dispatch FETCH_SIZE / WRITE_SIZE counters of tools/pmc_step_target.py (two rocprofv3 --pmc passes) to the
launches bench.py's roofline objects time, keyed by kernel + launch shape, and print the `per_launch_shape` block of
profiles/traffic.json.  HBM-side bytes = 2 * FETCH_SIZE KiB * 1024 + WRITE_SIZE KiB * 1024 (MI355X_MICROARCH.md, HBM section:
FETCH_SIZE reports half the bytes of 16-byte-per-lane streaming reads on gfx950; Infinity-Cache hits are counted).
usage: python3 tools/pmc_step_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <calls.json>"""
import collections
import csv
import json
import re
import sys

SYMS = {   # prof key -> (kernel symbols of one launch group, in dispatch order; '?' = optional follower, '1?' = at most one,
           # and only as the very next dispatch)
    'o2v_graph_fwd': ['o2v16_kernel<', '?o2v_combine_multi_kernel'],
    'o2v_graph_bwd': ['o2v16_bwd_scores_kernel<', 'o2v16_bwd_apply_kernel<', '?o2v_combine_multi_kernel'],
}
for _mode, _t in (('nt', 'false, false'), ('nn', 'false, true'), ('tn', 'true, true')):
    SYMS['gemm_f32_mfma_64x64_' + _mode] = ['gemm_kernel<64, 64, %s, 64>' % _t]
    SYMS['gemm_f32_mfma_128x64_' + _mode] = ['gemm_kernel_w3<128, 64, %s, 64>' % _t]
    SYMS['gemm_f32_mfma_128x128_' + _mode] = ['gemm_kernel_w3<128, 128, %s, 32>' % _t]
    SYMS['gemm_f32_mfma_256x256_' + _mode] = ['gemm_big_kernel<256, 256, %s>' % _t]
    SYMS['gemm_f32_mfma_streamk_256x256_' + _mode] = ['gemm_sk_kernel<%s>' % _t]
    # whole rounds on the 256 tile + the remaining rows on a smaller tile, one timed call (csrc/gemm.hip, gemm_plan)
    SYMS['gemm_f32_mfma_256x256+rest_' + _mode] = ['gemm_big_kernel<256, 256, %s>' % _t, '1?gemm_kernel']


def last_step(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
    a, b = adam[-2], adam[-1]
    return rows[a + 1:b + 1]


fetch, write, calls = last_step(sys.argv[1]), last_step(sys.argv[2]), json.load(open(sys.argv[3]))
# heads of the calls that ran as 'whole rounds on the 256 tile + the rest on a smaller tile'
rest_heads = [SYMS[c['key']][0] for c in calls if c['key'] in SYMS and '+rest' in c['key']]
out = collections.OrderedDict()
for key, syms in SYMS.items():
    mine = [c for c in calls if c['key'] == key]
    if not mine:
        continue
    per = []
    for rows, cname in ((fetch, 'FETCH_SIZE'), (write, 'WRITE_SIZE')):
        # walk the step: every dispatch of syms[0] opens a launch group, followers join it
        groups, cur, prev_head, prev_name = [], None, False, ''
        for r in rows:
            # (both tile heights of the stream-K kernel are one prof key)
            n = re.sub(r'gemm_sk_kernel<(\d+, )+', 'gemm_sk_kernel<', r['Kernel_Name'])
            was, prev_name = prev_name, n
            if '256' not in key and any(h in was for h in rest_heads):
                continue                    # the smaller-tile launch right behind a '+rest' head belongs to THAT call
            if syms[0] in n:
                cur = [float(r['Counter_Value'])]
                groups.append(cur)
                prev_head = True
                continue
            if cur is not None and any(s.startswith('1?') and s[2:] in n for s in syms[1:]):
                if prev_head:
                    cur.append(float(r['Counter_Value']))
                cur = None                  # only the dispatch right behind the head belongs to the call
            elif cur is not None and any(not s.startswith('1?') and s.lstrip('?') in n for s in syms[1:]):
                cur.append(float(r['Counter_Value']))
            elif cur is not None and key.startswith('o2v'):
                cur = None                  # a foreign kernel ends the group
            elif cur is not None and any(s.startswith('1?') for s in syms[1:]):
                cur = None
            prev_head = False
        per.append([sum(g) for g in groups])
    if len(per[0]) != len(mine) or len(per[1]) != len(mine):
        print('skip %s: %d logged launches, %d / %d dispatch groups' % (key, len(mine), len(per[0]), len(per[1])), file=sys.stderr)
        continue
    agg = collections.OrderedDict()
    for c, f, w in zip(mine, per[0], per[1]):
        d = agg.setdefault(key + ' | ' + c['shape'], {'launches': 0, 'FETCH_SIZE_KiB': 0.0, 'WRITE_SIZE_KiB': 0.0, 'work': c['algorithmic_work']})
        d['launches'] += 1; d['FETCH_SIZE_KiB'] += f; d['WRITE_SIZE_KiB'] += w
    for k, d in agg.items():
        n = d['launches']
        hb = (2 * d['FETCH_SIZE_KiB'] + d['WRITE_SIZE_KiB']) * 1024 / n
        ent = {'launches_in_step': n, 'FETCH_SIZE_KiB': round(d['FETCH_SIZE_KiB'] / n, 1), 'WRITE_SIZE_KiB': round(d['WRITE_SIZE_KiB'] / n, 1),
               'hbm_bytes_per_launch': int(hb)}
        if key.startswith('o2v'):
            ent['algorithmic_bytes'] = int(d['work'])
            ent['ratio'] = round(hb / d['work'], 2)
        out[k] = ent
print(json.dumps(out, indent=1))
