#!/bin/bash
# usage: tools/power_ab.sh <out.txt> [pairs]  -- VERDICT round 3 item 5: makes the "the step is power-coupled" claim checkable.
# Arms: default dispatch | DBX_WS_GATED=1 (ws kernel on the five gated 3x3 data gradients) | DBX_WS=0 (LDS band kernels everywhere) |
# zero data (DBX_BENCH_ZERO=1: same instruction streams, no operand toggling).  For every arm: <pairs> alternating runs of the un-instrumented
# training loop (ms/step), ONE kernel-trace pass and ONE --pmc pass (GRBM_GUI_ACTIVE) -> per-kernel us and effective GHz; rocm-smi power /
# clock samples while the default arm runs.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
out=$1; pairs=${2:-5}
B="python $R/bench.py --no-cpu-baseline --no-inference"
arms=("default:" "ws_gated:DBX_WS_GATED=1" "ws_off:DBX_WS=0" "zero_data:DBX_BENCH_ZERO=1")
: > $out
echo "# same box, one call; ms/step of the un-instrumented loop, $pairs alternating rounds" >> $out
for r in $(seq 1 $pairs); do
  for a in "${arms[@]}"; do
    name=${a%%:*}; envs=${a#*:}
    ms=$(env $envs $B --steps 30 --warmup 8 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "round $r $name $ms" >> $out
  done
done
python3 - $out <<'PY' >> $out
import sys, collections
d = collections.defaultdict(list)
for l in open(sys.argv[1]):
    p = l.split()
    if len(p) == 4 and p[0] == 'round': d[p[2]].append(float(p[3]))
for k, v in d.items(): print('# %-10s mean %.3f ms  min %.3f  max %.3f  (%d runs)' % (k, sum(v) / len(v), min(v), max(v), len(v)))
PY
for a in "${arms[@]}"; do
  name=${a%%:*}; envs=${a#*:}
  rm -rf /tmp/pa_kt /tmp/pa_pm
  env $envs rocprofv3 --kernel-trace -d /tmp/pa_kt -o k -- $B --steps 6 --warmup 3 > /tmp/pa_kt.log 2>&1
  env $envs rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d /tmp/pa_pm -o p -- $B --steps 2 --warmup 1 > /tmp/pa_pm.log 2>&1
  python3 $R/tools/rocpd_stats.py $(find /tmp/pa_kt -name "*.db" | head -1) > /tmp/pa_stats.txt 2>&1
  PMC_FILTER="conv|wgrad|head2" python3 $R/tools/pmc_summary.py $(find /tmp/pa_pm -name "*.db" | head -1) > /tmp/pa_pmc.txt 2>&1
  echo "" >> $out; echo "== arm $name ($envs): per-kernel average us (kernel trace), GRBM cycles (pmc pass), effective GHz, MFMA busy" >> $out
  python3 - <<'PY' >> $out
import re
dur = {}
for l in open('/tmp/pa_stats.txt'):
    p = l.split()
    if len(p) >= 7 and p[0].startswith('_Z'): dur[p[0]] = (int(p[1]), float(p[3]))
cur = None; d = {}
for l in open('/tmp/pa_pmc.txt'):
    if l and not l.startswith(' '): cur = l.strip(); d[cur] = {}
    elif l.strip():
        k, v = l.split(); d[cur][k] = float(v)
print('%-70s %6s %9s %10s %5s %5s' % ('kernel', 'calls', 'avg_us', 'cycles', 'GHz', 'busy'))
tot = 0.0
for k, (n, us) in sorted(dur.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    v = d.get(k) or d.get(re.sub(r'\.kd$', '', k)) or {}
    cyc = v.get('GRBM_GUI_ACTIVE')
    if us * n < 300: continue
    print('%-70s %6d %9.1f %10s %5s %5s' % (re.sub(r'^_Z\d+', '', k)[:70], n, us, '%.0f' % cyc if cyc else '-', '%.2f' % (cyc / us / 1e3) if cyc else '-',
          '%.2f' % (v['SQ_VALU_MFMA_BUSY_CYCLES'] / (32 * cyc)) if cyc and 'SQ_VALU_MFMA_BUSY_CYCLES' in v else '-'))
PY
done
echo "" >> $out; echo "== rocm-smi while the default arm runs (busiest samples)" >> $out
$B --steps 800 --warmup 20 > /tmp/pa_long.log 2>&1 &
BP=$!
: > /tmp/pa_smi.log
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "Package Power\|sclk\|junction" | sed 's/GPU\[0\]//; s/[[:space:]]\+/ /g' | tr '\n' '|' >> /tmp/pa_smi.log; echo >> /tmp/pa_smi.log
  sleep 0.2
done
grep -v "(94Mhz)\|(132Mhz)" /tmp/pa_smi.log | tail -12 >> $out
rocm-smi --showmaxpower 2>/dev/null | grep -i "Max" >> $out
cat $out | head -40
