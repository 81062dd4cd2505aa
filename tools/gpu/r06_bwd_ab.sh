#!/bin/bash
# round 6: the hand-scheduled data-gradient kernel (nsff_field_bwd_kernel_h3b) against the compiler-scheduled one (NSFF_BWD_KERNEL=c)
# on ONE box, interleaved: the kernels in isolation (tools/debug/bwd_bench.py: C2 fine-pass shape), then the whole training step.
#   usage: bash tools/gpu/r06_bwd_ab.sh <tag>
TAG=${1:-a}
O=gpurun_out/r06_$TAG; mkdir -p $O
for rnd in 1 2; do
  for k in h c; do
    echo "== round $rnd kernel $k (isolated, 196608 points)" >> $O/bwd_ab.txt
    NSFF_BWD_KERNEL=$k python tools/debug/bwd_bench.py 196608 20 2>&1 | grep -v amdgpu.ids >> $O/bwd_ab.txt
  done
done
for rnd in 1 2 3; do
  for k in h c; do
    echo "== round $rnd kernel $k (training step, eager)" >> $O/bwd_ab.txt
    NSFF_BWD_KERNEL=$k python bench.py --workload train --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms_per_step', round(d['ms_per_step'],4), 'value', round(d['value']/1e6,3), 'M ray-samples/s')" >> $O/bwd_ab.txt
  done
done
cat $O/bwd_ab.txt
