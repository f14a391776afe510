"""Idle time between the kernels of the replayed train step, from a rocprofv3 --kernel-trace CSV of bench.py:
for the last `steps` steps (found by their Adam launch) prints step wall time, summed kernel time, summed idle gaps on the
busiest stream and the gap histogram.  usage: python tools/archive/trace_gaps.py <kernel_trace.csv> [steps=3]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows))
adam = [i for i, e in enumerate(ev) if 'adam_kernel' in e[2]]
# steps = spans between consecutive Adam launches of the timed region (the last ones in the trace are the bench's other legs)
spans = [(adam[i], adam[i + 1]) for i in range(len(adam) - 1)]
# keep spans with the most common kernel count (the batch-64 fp32 graph step)
cnt = collections.Counter(b - a for a, b in spans)
n0 = cnt.most_common(1)[0][0]
spans = [s for s in spans if s[1] - s[0] == n0][:steps]
for a, b in spans:
    seg = ev[a + 1:b + 1]
    wall = seg[-1][1] - ev[a][1]
    busy = 0
    gaps = []
    cur_end = ev[a][1]
    for s, e, _ in seg:
        if s > cur_end:
            gaps.append(s - cur_end)
        busy += e - s
        cur_end = max(cur_end, e)
    hist = collections.Counter(min(g // 1000, 10) for g in gaps)
    print('kernels %d  wall %.3f ms  sum of kernel durations %.3f ms  idle %.3f ms in %d gaps  (gap histogram us: %s)' %
          (len(seg), wall / 1e6, busy / 1e6, sum(gaps) / 1e6, len(gaps), dict(sorted(hist.items()))))
big = sorted(((ev[i + 1][0] - ev[i][1], ev[i][2][:60], ev[i + 1][2][:60]) for i in range(spans[0][0], spans[0][1])), reverse=True)[:12]
for g, x, y in big:
    print('%7.1f us  after %-60s before %s' % (g / 1e3, x, y))
