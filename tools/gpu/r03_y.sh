#!/bin/bash
# call Y: inference epilogue split around the barrier (conversions before, LDS stores after)
cd $GRAFT_REPO_ROOT
for v in _noepi "" _noepi ""; do
  echo "== lib$v"
  NSFF_LIB=$GRAFT_REPO_ROOT/nsff_pl_amd/libnsff_hip$v.so timeout 300 python tools/bench_field.py 2>&1 | tail -1
  NSFF_LIB=$GRAFT_REPO_ROOT/nsff_pl_amd/libnsff_hip$v.so NSFF_TILE_POINTS=64 timeout 300 python tools/bench_field.py 2>&1 | tail -1
done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
