"""Kernel timeline of one train step from a rocprofv3 results .db (the default output format when no CSV is asked for):
usage: python tools/db_timeline.py <t_results.db> [first_us] [last_us]   -- start (us from the step's first kernel), duration,
gap to the previous kernel's end, stream, name."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = list(cur.execute(f'select d.start, d.end, d.stream_id, s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id order by d.start'))
idx = [i for i, r in enumerate(rows) if 'melspec_r16' in r[3]]
a, b = idx[-2], idx[-1]
lo = float(sys.argv[2]) if len(sys.argv) > 2 else -1e9
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
t0, pe = rows[a][0], rows[a][0]
print(f'step {(rows[b][0] - t0) / 1e3:.1f} us, {b - a} launches')
for s, e, st, n in rows[a:b + 1]:
    n = n.replace('nafp::', '').replace('void ', '')[:64]
    t = (s - t0) / 1e3
    if lo <= t <= hi:
        print(f'{t:9.1f} {(e - s) / 1e3:8.1f} gap{(s - pe) / 1e3:7.1f} s{st} {n}')
    pe = max(pe, e)
