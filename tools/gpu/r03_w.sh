#!/bin/bash
# call W: splat tile pass with four samples in flight per thread + 12 KiB histogram (three workgroups per CU)
cd $GRAFT_REPO_ROOT
for v in _x0 ""; do
  echo "== lib$v"
  NSFF_LIB=$GRAFT_REPO_ROOT/nsff_pl_amd/libnsff_hip$v.so timeout 300 python tools/bench_interp.py --planes 256 --reps 10 2>&1 | tail -1
  NSFF_LIB=$GRAFT_REPO_ROOT/nsff_pl_amd/libnsff_hip$v.so timeout 300 python tools/bench_interp.py --planes 256 --reps 10 --flow 0.2 2>&1 | tail -1
done
timeout 900 python -m pytest tests/test_interpolate.py -m gpu -x -q 2>&1 | tail -3
