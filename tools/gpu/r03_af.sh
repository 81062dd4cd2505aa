#!/bin/bash
# call AF: eight k-steps of weights in flight (inference, 32-neuron tiling) vs four
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
for v in _ring4 "" _ring4 ""; do
  echo "== lib$v"
  NSFF_LIB=$GRAFT_REPO_ROOT/nsff_pl_amd/libnsff_hip$v.so timeout 300 python tools/bench_field.py 2>&1 | tail -1
done
NSFF_LIB=$GRAFT_REPO_ROOT/nsff_pl_amd/libnsff_hip_timing.so timeout 300 python tools/debug/h3_timing.py 0 f16x3 2>&1 | tail -11 | cut -c1-230
