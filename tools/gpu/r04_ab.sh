#!/bin/bash
TAG=${1:-ab}; mkdir -p gpurun_out/r04_$TAG
for rep in 1 2; do
for v in 1 0; do
  NSFF_RNG_SIDE=$v python bench.py --no-aux --no-cpu-baseline --steps 100 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('rng_side=$v', 'ms/step %.4f median %.4f field launch %.4f ms clock %.3f -> non-field per step %.4f ms' % (d['ms_per_step'], d['ms_per_step_median_events'], r['avg_launch_ms'], r['clock_ghz'] or 0, d['ms_per_step'] - 4 * r['avg_launch_ms']))"
done; done | tee gpurun_out/r04_$TAG/ab.txt
