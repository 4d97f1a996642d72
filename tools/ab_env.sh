#!/bin/bash
# usage: tools/ab_env.sh <rounds> <steps> VAR=a VAR=b [...] -- same-box interleaved A/B of the whole training step between settings of ONE
# environment switch of the product library (e.g. DBX_SGD_PACK=1 DBX_SGD_PACK=0); prints ms/step per run and the means
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
rounds=$1; steps=$2; shift; shift
for r in $(seq 1 $rounds); do
  for kv in "$@"; do
    ms=$(env $kv python $R/bench.py --no-cpu-baseline --no-inference --steps $steps --warmup 8 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "round $r $kv $ms"
  done
done | tee /tmp/ab_env.txt
python3 - <<'PY'
import collections
d=collections.defaultdict(list)
for l in open('/tmp/ab_env.txt'):
    p=l.split(); d[p[2]].append(float(p[3]))
for k,v in d.items(): print('%-24s mean %.3f ms  min %.3f  runs %s' % (k, sum(v)/len(v), min(v), v))
PY
