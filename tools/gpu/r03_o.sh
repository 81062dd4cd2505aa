#!/bin/bash
# call O: fast mode on the eight-wave 128-point tiling (no spills, conflict-free encoder mapping) vs the four-wave form
cd $GRAFT_REPO_ROOT
for v in "" _fast8w; do
  echo "== lib$v"
  NSFF_LIB=$GRAFT_REPO_ROOT/nsff_pl_amd/libnsff_hip$v.so timeout 300 python tools/bench_field.py --precision f16 2>&1 | tail -4
done
NSFF_LIB=$GRAFT_REPO_ROOT/nsff_pl_amd/libnsff_hip_fast8w.so timeout 600 python -m pytest tests/test_fast_mode.py -m gpu -x -q 2>&1 | tail -4
