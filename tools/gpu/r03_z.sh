#!/bin/bash
# call Z: s_setprio for one wave of each SIMD during the GEMM (alone / with the epilogue split around the barrier)
cd $GRAFT_REPO_ROOT
for v in "" _p1 _p1e "" _p1 _p1e; do
  echo "== lib$v"
  NSFF_LIB=$GRAFT_REPO_ROOT/nsff_pl_amd/libnsff_hip$v.so timeout 300 python tools/bench_field.py 2>&1 | tail -1
done
NSFF_LIB=$GRAFT_REPO_ROOT/nsff_pl_amd/libnsff_hip_p1e.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "f16x3" 2>&1 | tail -2
