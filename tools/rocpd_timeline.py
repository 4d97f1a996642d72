"""Kernel sequence of the LAST training step in a rocprofv3 rocpd database (kernel trace): start offset, duration, idle gap
before the launch, grid size, name.  A step is delimited by consecutive loss_kernel launches.  usage: rocpd_timeline.py <db>"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
cols = [r[1] for r in cur.execute('pragma table_info(%s)' % kd)]
scol = [r[1] for r in cur.execute('pragma table_info(%s)' % ks)]
namecol = 'kernel_name' if 'kernel_name' in scol else 'display_name'
gx = [c for c in cols if 'grid' in c]
rows = list(cur.execute('select s.%s, d.start, d.end, %s from %s d join %s s on d.kernel_id = s.id order by d.start' % (namecol, ','.join('d.' + c for c in gx), kd, ks)))
loss = [i for i, r in enumerate(rows) if 'loss_kernel' in r[0]]
if len(loss) < 3: print('need >= 3 steps'); sys.exit(1)
a, b = loss[-3], loss[-2]                 # one full period loss -> loss (backward of step k, forward of step k+1)
t0 = rows[a][1]; prev = None; busy = 0
print('cols:', gx)
for r in rows[a:b]:
    name = re.sub(r'\(.*\)$', '', r[0]); name = re.sub(r'^_Z\d+', '', name)[:60]
    gap = 0 if prev is None else (r[1] - prev) / 1e3
    busy += (r[2] - r[1]) / 1e3
    print('%9.1f %8.1f gap %7.1f grid %-22s %s' % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, gap, 'x'.join(str(v) for v in r[3:]), name))
    prev = max(prev or 0, r[2])
print('period %.1f us, kernel busy sum %.1f us' % ((rows[b][1] - t0) / 1e3, busy))
