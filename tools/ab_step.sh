#!/bin/bash
# usage: tools/ab_step.sh <rounds> [steps] -- same-box interleaved A/B of the whole training step: the product library and every
# library under densebox_amd/csrc/variants/ run the bench's training loop alternately; prints ms/step per run and the means
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
rounds=${1:-3}; steps=${2:-30}
for r in $(seq 1 $rounds); do
  for lib in product $(ls $R/densebox_amd/csrc/variants/*.so 2>/dev/null); do
    tag=$(basename $lib .so | sed 's/libdensebox_hip_//')
    if [ "$lib" == "product" ]; then unset DBX_LIB; else export DBX_LIB=$lib; fi
    ms=$(python $R/bench.py --no-cpu-baseline --no-inference --steps $steps --warmup 8 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "round $r $tag $ms"
  done
done | tee /tmp/ab_step.txt
python3 - <<'PY'
import collections
d=collections.defaultdict(list)
for l in open('/tmp/ab_step.txt'):
    p=l.split(); d[p[2]].append(float(p[3]))
for k,v in d.items(): print('%-16s mean %.3f ms  min %.3f  runs %s' % (k, sum(v)/len(v), min(v), v))
PY
