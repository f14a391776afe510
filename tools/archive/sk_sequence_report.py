"""Group the region-projection dispatches of tools/archive/sk_sequence_probe.py by sequence.  usage: python3 tools/archive/sk_sequence_report.py <kernel_trace.csv>"""
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = {1: 'A back to back', 2: 'B behind a 0.5-ms memory-bound update', 3: 'C behind 100 light kernels', 4: 'D behind the deep TN launch',
         5: 'E after 2 ms of idle'}
seq, run, out = 0, 0, {}
for r in rows:
    n = r['Kernel_Name']
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if 'FillFunctor' in n and int(r.get('Grid_Size', r.get('Grid_Size_X', '0')) or 0) <= 4096 and d < 20 and 'gemm' not in n:
        run += 1
        continue
    if run:
        seq, run = run, 0
    if ('gemm_sk_kernelILi256ELb0ELb0' in n or 'gemm_sk_kernel<256, false, false>' in n) and seq in names:
        out.setdefault(names[seq], []).append(d)
res = {k: {'n': len(v), 'median_us': round(sorted(v)[len(v) // 2], 1), 'min_us': round(min(v), 1), 'max_us': round(max(v), 1)} for k, v in out.items()}
print(json.dumps(res, indent=1))
