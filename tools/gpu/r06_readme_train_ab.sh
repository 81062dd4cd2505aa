#!/bin/bash
# round 6: the README training configuration's step with the view-direction static trunk's training forward on the hand-scheduled
# SAVE body (default) against the eight-wave SAVE kernel (NSFF_NO_SIDE_BIAS=1), and the data-gradient kernels (NSFF_BWD_KERNEL=c):
# interleaved on ONE box.   usage: bash tools/gpu/r06_readme_train_ab.sh <tag>
TAG=${1:-a}
O=gpurun_out/r06_$TAG; mkdir -p $O
for rnd in 1 2 3; do
  for v in "default" "NSFF_NO_SIDE_BIAS=1" "NSFF_BWD_KERNEL=c" "NSFF_NO_SIDE_BIAS=1 NSFF_BWD_KERNEL=c"; do
    echo "== round $rnd: $v" >> $O/readme_train_ab.txt
    if [ "$v" = "default" ]; then python tools/debug/readme_train_timing.py 20 2>/dev/null | grep -E "ms_per_step|kernels|h3|\"c" | tr -d '\n' >> $O/readme_train_ab.txt
    else env $v python tools/debug/readme_train_timing.py 20 2>/dev/null | grep -E "ms_per_step|kernels|h3|\"c" | tr -d '\n' >> $O/readme_train_ab.txt; fi
    echo >> $O/readme_train_ab.txt
  done
done
cat $O/readme_train_ab.txt
