"""Idle time between kernels in a rocprofv3 (rocpd / SQLite) trace: python3 tools/archive/rocpd_gaps.py <results.db> [skip_first_ms=0] [length_ms]
(skip_first_ms < 0: counted back from the end of the trace)
Prints the busy time (union of kernel intervals), the wall time of the traced span and the largest gaps with the kernels around them."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
skip = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 0.0
length = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else None
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'info_kernel_symbol' in t][0]
rows = cur.execute("select d.start, d.end, s.kernel_name from %s d join %s s on d.kernel_id=s.id order by d.start" % (kd, ks)).fetchall()
t0 = rows[0][0]
if skip < 0:
    skip = rows[-1][1] - t0 + skip
rows = [r for r in rows if r[0] - t0 >= skip and (length is None or r[0] - t0 < skip + length)]
busy, end, gaps = 0, rows[0][0], []
prev = None
for st, en, name in rows:
    if st > end:
        gaps.append((st - end, prev, name, st - t0))
        busy += en - st
        end = en
    elif en > end:
        busy += en - end
        end = en
    prev = name
wall = end - rows[0][0]
short = lambda n: re.sub(r'\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+', '', n or '')[:50]
print('kernels %d  wall %.3f ms  busy %.3f ms  idle %.3f ms (%.1f %%)' % (len(rows), wall / 1e6, busy / 1e6, (wall - busy) / 1e6, 100.0 * (wall - busy) / wall))
hist = {}
for g, a, b, at in gaps:
    k = '<5us' if g < 5e3 else '<20us' if g < 2e4 else '<100us' if g < 1e5 else '<1ms' if g < 1e6 else '>=1ms'
    hist[k] = hist.get(k, [0, 0])
    hist[k][0] += 1; hist[k][1] += g
print('gap histogram (count, total ms):', {k: (v[0], round(v[1] / 1e6, 3)) for k, v in hist.items()})
for g, a, b, at in sorted(gaps, reverse=True)[:25]:
    print('%9.1f us at %10.3f ms  after %-50s before %s' % (g / 1e3, at / 1e6, short(a), short(b)))
