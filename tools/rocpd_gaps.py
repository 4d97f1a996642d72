"""Median idle gap in front of every launch of tools/probe_boundary under rocprofv3, grouped by (grid, block, LDS) of the launched kernel.
usage: rocpd_gaps.py <db>"""
import sqlite3, sys, statistics
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
cols = [r[1] for r in cur.execute('pragma table_info(%s)' % kd)]
want = [c for c in cols if 'grid_size_x' in c or 'workgroup_size_x' in c or 'lds' in c.lower()]
rows = list(cur.execute('select start, end, %s from %s order by start' % (','.join(want), kd)))
groups = {}
order = []
for i in range(1, len(rows)):
    key = tuple(rows[i][2:])
    if key not in groups:
        order.append(key)
    groups.setdefault(key, []).append((rows[i][0] - rows[i - 1][1]) / 1e3)
print('columns:', want)
# (the last section of the probe: pairs of a (512, 131072) and a (256, 65536) launch per host call, 30 us of host work between calls)
for k in order:
    g = groups[k]
    print('%-40s launches %5d  gap median %7.2f us  min %7.2f  max %7.2f' % (str(k), len(g), statistics.median(g), min(g), max(g)))
