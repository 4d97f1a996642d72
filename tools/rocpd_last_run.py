"""Kernel sequence of the LAST repetition in a rocprofv3 kernel trace of a repeated launch sequence (period = distance between
the last two launches of the named marker kernel).  usage: rocpd_last_run.py <db> <marker substring>"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor(); mark = sys.argv[2]
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
cols = [r[1] for r in cur.execute('pragma table_info(%s)' % kd)]
scol = [r[1] for r in cur.execute('pragma table_info(%s)' % ks)]
namecol = 'kernel_name' if 'kernel_name' in scol else 'display_name'
rows = list(cur.execute('select s.%s, d.start, d.end, d.grid_size_x, d.workgroup_size_x from %s d join %s s on d.kernel_id = s.id order by d.start' % (namecol, kd, ks)))
m = [i for i, r in enumerate(rows) if mark in r[0]]
a, b = m[-2], m[-1]
t0 = rows[a][1]; prev = None; busy = 0
for r in rows[a:b]:
    name = re.sub(r'\(.*\)$', '', r[0]); name = re.sub(r'^_Z\d+', '', name)[:70]
    gap = 0 if prev is None else (r[1] - prev) / 1e3
    busy += (r[2] - r[1]) / 1e3
    print('%8.1f %7.1f gap %5.1f wgs %6d  %s' % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, gap, r[3] // max(r[4], 1), name))
    prev = max(prev or 0, r[2])
print('period %.1f us, kernel busy sum %.1f us' % ((rows[b][1] - t0) / 1e3, busy))
