#!/bin/bash
# round 5: the persistent launch of nsff_field_kernel_h3a (H3AArgs::p_mode) -- parity files, then interleaved same-box A/B of the
# headline workload with and without it (NSFF_NO_PERSIST=1 = one workgroup per tile).  usage: bash tools/gpu/r05_persist_ab.sh <tag>
TAG=${1:-p}; O=gpurun_out/r05_$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_h3a_cross.py tests/test_graphs.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
for i in 1 2 3; do
  for mode in persist tile; do
    if [ $mode = tile ]; then export NSFF_NO_PERSIST=1; else unset NSFF_NO_PERSIST; fi
    timeout 300 python bench.py --no-cpu-baseline --no-aux --steps 100 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']
print('$mode', $i, d['value'], d['ms_per_step'], r.get('frac'), r.get('frac_at_clock'), r.get('clock_mhz'))" | tee -a $O/ab.txt
  done
done
for w in eval eval_interp; do
  for mode in persist tile; do
    if [ $mode = tile ]; then export NSFF_NO_PERSIST=1; else unset NSFF_NO_PERSIST; fi
    timeout 300 python bench.py --workload $w --no-cpu-baseline --no-aux 2>/dev/null | tail -1 | cut -c1-400 | sed "s/^/$w $mode /" | tee -a $O/ab.txt
  done
done
