"""Timeline of ONE replayed train step from a rocprofv3 (rocpd / SQLite) kernel trace of tools/profile_step.py ... graphs:
python3 tools/rocpd_timeline.py <results.db> [step=-2] [sequence.txt]

The step is cut at its Adam launches (one per step); inside it the kernels are put into phases by name (encoder forward, word
loop forward, loss + head, word loop backward, encoder backward, Adam) and for each phase the tool prints its wall time, the
sum of its kernels' durations, the time during which NO kernel of the device was running (gaps between dependent launches of
the replayed graph) and the time during which two or more were.  Writes JSON to stdout."""
import json
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'info_kernel_symbol' in t][0]
rows = cur.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id=s.id order by d.start" % (kd, ks)).fetchall()
rows = [(re.sub(r'\(anonymous namespace\)::', '', n), a, b) for n, a, b in rows]
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r[0]]
# a step = the kernels after the previous step's last Adam launch up to and including this step's last Adam launch
ends = [i for j, i in enumerate(adam) if j + 1 == len(adam) or adam[j + 1] - i > 50]
lo, hi = ends[which - 1] + 1, ends[which] + 1
step = rows[lo:hi]
t0 = step[0][1]


def short(n):
    m = re.match(r'(?:void )?([A-Za-z0-9_]+)', n)
    return m.group(1) if m else n[:40]


WORD_F = ('dec_mid_fwd', 'dec_tail_fwd', 'select_embed')
WORD_B = ('dec_mid_bwd', 'lstm_pw_bwd')
first_wf = next(i for i, r in enumerate(step) if any(k in r[0] for k in WORD_F))
last_wf = max(i for i, r in enumerate(step) if any(k in r[0] for k in WORD_F))
first_wb = next(i for i, r in enumerate(step) if any(k in r[0] for k in WORD_B))
last_wb = max(i for i, r in enumerate(step) if any(k in r[0] for k in WORD_B))
# the skinny launch in front of the first dec_mid_fwd belongs to the loop
bounds = [('encoder forward + decoder caches', 0, first_wf - 1), ('word loop forward', first_wf - 1, last_wf + 1),
          ('vocabulary head, loss, head backward', last_wf + 1, first_wb), ('word loop backward', first_wb, last_wb + 1),
          ('decoder weight gradients, encoder backward, Adam', last_wb + 1, len(step))]


def cover(ks_):
    """(time covered by >= 1 kernel, time covered by >= 2) of a list of (name, start, end)"""
    ev = []
    for _, a, b in ks_:
        ev.append((a, 1))
        ev.append((b, -1))
    ev.sort()
    c1 = c2 = 0
    depth = 0
    prev = None
    for t, d in ev:
        if prev is not None:
            if depth >= 1:
                c1 += t - prev
            if depth >= 2:
                c2 += t - prev
        depth += d
        prev = t
    return c1, c2


out = {'step_wall_us': round((max(r[2] for r in step) - t0) / 1e3, 1), 'kernels': len(step), 'phases': []}
for name, a, b in bounds:
    part = step[a:b]
    if not part:
        continue
    w0, w1 = part[0][1], max(r[2] for r in part)
    if b < len(step):
        w1 = max(w1, step[b][1]) if step[b][1] > w1 else w1
    # everything of the device running inside the phase's window (side-stream launches of other phases included)
    inside = [(n, max(s, w0), min(e, w1)) for n, s, e in step if e > w0 and s < w1]
    c1, c2 = cover(inside)
    by = {}
    for n, s, e in part:
        k = short(n)
        by.setdefault(k, [0, 0.0])
        by[k][0] += 1
        by[k][1] += (e - s) / 1e3
    top = sorted(by.items(), key=lambda kv: -kv[1][1])[:8]
    out['phases'].append({'phase': name, 'launches': len(part), 'wall_us': round((w1 - w0) / 1e3, 1),
                          'sum_of_kernels_us': round(sum(e - s for _, s, e in part) / 1e3, 1),
                          'device_idle_us': round((w1 - w0 - c1) / 1e3, 1), 'two_or_more_running_us': round(c2 / 1e3, 1),
                          'top': [{'kernel': k, 'calls': v[0], 'us': round(v[1], 1)} for k, v in top]})
if len(sys.argv) > 3:
    # the launch sequence of the step: offset from its start (us), duration (us), kernel
    with open(sys.argv[3], 'w') as f:
        for n, a, b in step:
            f.write('%9.1f %8.1f  %s\n' % ((a - t0) / 1e3, (b - a) / 1e3, short(n)))
print(json.dumps(out, indent=1))
