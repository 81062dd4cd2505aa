#!/bin/bash
# round 5: what the part does under the field kernel -- rocm-smi power / clocks sampled while bench.py loops (gpurun_out/r05_power.txt)
O=gpurun_out/r05_power.txt
rocm-smi --showpower --showmaxpower --showclocks --showperflevel > $O 2>&1
python bench.py --steps 20000 --warmup 20 --no-cpu-baseline --no-aux > gpurun_out/r05_power_bench.json 2>/dev/null &
PID=$!
sleep 16
for i in 1 2 3 4 5 6; do
  echo "---- sample $i (bench.py running) ----" >> $O
  rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -E "Power|sclk|mclk|fclk|Temperature|socclk" >> $O
  sleep 1
done
wait $PID
tail -1 gpurun_out/r05_power_bench.json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('bench during the samples:', d['value'], d['ms_per_step'], 'clock_ghz', r.get('clock_ghz'), 'frac', r['frac'])" >> $O
cat $O
