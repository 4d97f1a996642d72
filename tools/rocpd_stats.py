"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) into a per-kernel stats table (like --stats CSV)."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
cols = [r[1] for r in cur.execute('pragma table_info(%s)' % kd)]
scol = [r[1] for r in cur.execute('pragma table_info(%s)' % ks)]
namecol = 'kernel_name' if 'kernel_name' in scol else ('display_name' if 'display_name' in scol else 'name')
q = 'select s.%s, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from %s d join %s s on d.kernel_id = s.id group by s.%s order by 3 desc' % (namecol, kd, ks, namecol)
rows = list(cur.execute(q))
tot = sum(r[2] for r in rows)
print('%-110s %7s %12s %10s %10s %10s %6s' % ('kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', '%'))
for name, n, t, mn, mx in rows:
    name = re.sub(r'\(.*\)$', '', name)
    print('%-110s %7d %12.1f %10.2f %10.2f %10.2f %6.2f' % (name[:110], n, t / 1e3, t / n / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot))
print('TOTAL kernel time us: %.1f' % (tot / 1e3))
