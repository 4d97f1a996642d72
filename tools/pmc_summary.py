import sqlite3, sys, re, collections, os
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
def T(p): return [t for t in tabs if t.startswith(p)][0]
pe, pmc, kd, ks = T('rocpd_pmc_event'), T('rocpd_info_pmc'), T('rocpd_kernel_dispatch'), T('rocpd_info_kernel_symbol')
scol = [r[1] for r in cur.execute('pragma table_info(%s)' % ks)]
namecol = 'kernel_name' if 'kernel_name' in scol else 'display_name'
pcols = [r[1] for r in cur.execute('pragma table_info(%s)' % pe)]
q = 'select s.%s, p.name, avg(e.value), count(*) from %s e join %s p on e.pmc_id = p.id join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id group by 1,2' % (namecol, pe, pmc, kd, ks)
try:
    rows = list(cur.execute(q))
except Exception as ex:
    print('query failed', ex, pcols); sys.exit(1)
by = collections.defaultdict(dict)
for k, n, v, c in rows: by[re.sub(r'\(.*\)$', '', k)][n] = v
for k, d in by.items():
    if (re.search(os.environ['PMC_FILTER'], k) if os.environ.get('PMC_FILTER') else ('conv' in k or 'wgrad' in k)):
        print(k[:100]); [print('    %-32s %16.1f' % (n, v)) for n, v in sorted(d.items())]
