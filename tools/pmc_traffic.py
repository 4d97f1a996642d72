"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; units of KiB per the counter definition).
usage: python tools/pmc_traffic.py <fetch_results.db> <write_results.db> [out.json]
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half of the bytes of wide (16 B/lane) coalesced reads, so it
is doubled; WRITE_SIZE is taken as is (uncalibrated)."""
import sqlite3, sys, re, json, collections

def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    T = lambda p: [t for t in tabs if t.startswith(p)][0]
    pe, pmc, kd, ks = T('rocpd_pmc_event'), T('rocpd_info_pmc'), T('rocpd_kernel_dispatch'), T('rocpd_info_kernel_symbol')
    q = ('select s.kernel_name, sum(e.value), count(*) from %s e join %s p on e.pmc_id = p.id join %s d on e.event_id = d.event_id '
         'join %s s on d.kernel_id = s.id where p.name = ? group by 1' % (pe, pmc, kd, ks))
    return {re.sub(r'\(.*\)$', '', k).replace('.kd', ''): (v, n) for k, v, n in cur.execute(q, (counter,))}

f = per_kernel(sys.argv[1], 'FETCH_SIZE'); w = per_kernel(sys.argv[2], 'WRITE_SIZE')
out = {}
print('%-100s %8s %14s %14s %14s' % ('kernel', 'launches', 'fetch MB/launch', 'write MB/launch', 'HBM MB/launch'))
for k in sorted(f, key=lambda k: -(2 * f[k][0] + w.get(k, (0, 1))[0])):
    fv, n = f[k]; wv, wn = w.get(k, (0.0, n))
    fetch = 2.0 * fv * 1024 / n; write = wv * 1024 / max(wn, 1)
    out[k] = {'launches': n, 'fetch_bytes_per_launch': fetch, 'write_bytes_per_launch': write, 'hbm_bytes_per_launch': fetch + write}
    print('%-100s %8d %14.1f %14.1f %14.1f' % (k[:100], n, fetch / 1e6, write / 1e6, (fetch + write) / 1e6))
if len(sys.argv) > 3:
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import csrc_hash
    json.dump({'csrc_hash': csrc_hash(), 'kernels': out}, open(sys.argv[3], 'w'), indent=1)
