# usage: tools/power_sample.sh  -- samples rocm-smi power / clocks while a long bench run is in flight; keeps the busiest samples
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
mkdir -p $R/gpurun_out/pw
python $R/bench.py --no-cpu-baseline --no-inference --steps 1500 --warmup 20 > $R/gpurun_out/pw/bench.log 2>&1 &
BP=$!
: > $R/gpurun_out/pw/all.log
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "Package Power\|sclk\|junction" | sed 's/GPU\[0\]//; s/[[:space:]]\+/ /g' | tr '\n' '|' >> $R/gpurun_out/pw/all.log; echo >> $R/gpurun_out/pw/all.log
  sleep 0.2
done
grep -v "(94Mhz)\|(132Mhz)" $R/gpurun_out/pw/all.log | tail -30 > $R/gpurun_out/pw/smi.log
tail -1 $R/gpurun_out/pw/bench.log | grep -o '"ms_per_step": [0-9.]*' >> $R/gpurun_out/pw/smi.log
rocm-smi --showmaxpower 2>/dev/null | grep -i "Max" >> $R/gpurun_out/pw/smi.log
