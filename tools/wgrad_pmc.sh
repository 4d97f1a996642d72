#!/bin/bash
# usage: tools/wgrad_pmc.sh <tag> <layers> [env...]  -- LDS / wait counters of the weight-gradient kernels on the lab shapes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
tag=$1; layers=$2; shift; shift
rm -rf /tmp/wp_$tag
env "$@" rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d /tmp/wp_$tag -o p -- python $R/tools/gpu_wgrad_bench.py bf16 3 $layers > /tmp/wp_$tag.log 2>&1
python3 - $(find /tmp/wp_$tag -name "*.db" | head -1) > $R/gpurun_out/${tag}_lds.txt 2>&1 <<'PY'
import sqlite3, sys, re, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
T = lambda p: [t for t in tabs if t.startswith(p)][0]
pe, pmc, kd, ks = T('rocpd_pmc_event'), T('rocpd_info_pmc'), T('rocpd_kernel_dispatch'), T('rocpd_info_kernel_symbol')
q = 'select s.kernel_name, p.name, avg(e.value) from %s e join %s p on e.pmc_id = p.id join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id group by 1,2' % (pe, pmc, kd, ks)
by = collections.defaultdict(dict)
for k, n, v in cur.execute(q): by[re.sub(r'\(.*\)$', '', k)][n] = v
for k, d in by.items():
    if 'wgrad' in k and 'reduce' not in k:
        print(k[:60]); [print('    %-28s %14.0f' % (n, v)) for n, v in sorted(d.items())]
PY
cat $R/gpurun_out/${tag}_lds.txt
