"""Per-kernel totals from a rocprofv3 (rocpd / SQLite) result database: python3 tools/rocpd_stats.py <results.db> [top=40] [divide=1]
`divide`: report per-unit figures (e.g. the number of critic updates the target ran).  Writes CSV to stdout."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
div = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'info_kernel_symbol' in t][0]
rows = cur.execute("select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start) from %s d join %s s on d.kernel_id=s.id "
                   "group by s.kernel_name order by 3 desc" % (kd, ks)).fetchall()
tot = sum(r[2] for r in rows)
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","CallsPerUnit","UsPerUnit"')
print('"TOTAL",%d,%d,,100.0,%.1f,%.1f' % (sum(r[1] for r in rows), tot, sum(r[1] for r in rows) / div, tot / div / 1e3))
for n, c, t, a in rows[:top]:
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    print('"%s",%d,%d,%.0f,%.2f,%.1f,%.1f' % (n, c, t, a, 100.0 * t / tot, c / div, t / div / 1e3))
