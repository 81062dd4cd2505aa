#!/bin/bash
# round 6: the PERSISTENT form of nsff_field_bwd_kernel_h3b (one workgroup per compute unit, items from a device counter, the next
# item's records DMA'd into LDS behind the current body) against one workgroup per item (NSFF_BWD_PERSIST=0), ONE box, interleaved:
# the kernel in isolation (tools/debug/bwd_bench.py: C2 fine-pass shape), then the whole training step.
#   usage: bash tools/gpu/r06_bwd_persist_ab.sh <tag>
TAG=${1:-a}
O=gpurun_out/r06_$TAG; mkdir -p $O
F=$O/bwd_persist_ab.txt
for rnd in 1 2 3; do
  for p in 1 0; do
    echo "== round $rnd NSFF_BWD_PERSIST=$p (isolated, 196608 points)" >> $F
    NSFF_BWD_PERSIST=$p python tools/debug/bwd_bench.py 196608 20 2>&1 | grep "field_backward" | sed 's/.*| field_backward/  field_backward/' >> $F
  done
done
for rnd in 1 2 3; do
  for p in 1 0; do
    echo "== round $rnd NSFF_BWD_PERSIST=$p (training step, eager)" >> $F
    NSFF_BWD_PERSIST=$p python bench.py --workload train --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms_per_step', round(d['ms_per_step'],4), 'value', round(d['value']/1e6,3), 'M ray-samples/s')" >> $F
  done
done
cat $F
