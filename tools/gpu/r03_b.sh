#!/bin/bash
# round 3, call B: phase timing (s_memtime stamps) of the forward (inference default tiling, training forward) and backward kernels
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_b; mkdir -p $O
export NSFF_LIB=nsff_pl_amd/libnsff_hip_timing.so
for args in "0 f16x3" "64 f16x3" "0 f16x3 save" "64 f16x3 save"; do
  echo "==== h3_timing $args" >> $O/timing.txt
  timeout 300 python tools/debug/h3_timing.py $args >> $O/timing.txt 2>&1
done
echo "==== bwd_timing" >> $O/timing.txt
timeout 300 python tools/debug/bwd_timing.py >> $O/timing.txt 2>&1
unset NSFF_LIB
timeout 900 python -m pytest tests/test_dist_gpu.py -m gpu -x -q > $O/pytest_dist.log 2>&1; echo "pytest dist rc=$?" >> $O/timing.txt
tail -15 $O/pytest_dist.log >> $O/timing.txt
cat $O/timing.txt
