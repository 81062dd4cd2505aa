#!/bin/bash
# call AD: progress-based fair priority between the two waves that share a SIMD's matrix pipe
cd $GRAFT_REPO_ROOT
for v in _nofair "" _nofair ""; do
  echo "== lib$v"
  NSFF_LIB=$GRAFT_REPO_ROOT/nsff_pl_amd/libnsff_hip$v.so timeout 300 python tools/bench_field.py 2>&1 | tail -1
done
NSFF_LIB=$GRAFT_REPO_ROOT/nsff_pl_amd/libnsff_hip_timing.so timeout 300 python tools/debug/h3_timing.py 0 f16x3 2>&1 | tail -12 | cut -c1-260
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
