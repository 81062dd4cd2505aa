#!/bin/bash
# call U: splat tile pass with fewer planes per workgroup (more workgroups per CU)
cd $GRAFT_REPO_ROOT
for v in "" _sb _sa _sc; do
  echo "== lib$v"
  NSFF_LIB=$GRAFT_REPO_ROOT/nsff_pl_amd/libnsff_hip$v.so timeout 300 python tools/bench_interp.py --planes 256 --reps 10 2>&1 | tail -2
  NSFF_LIB=$GRAFT_REPO_ROOT/nsff_pl_amd/libnsff_hip$v.so timeout 300 python tools/bench_interp.py --planes 256 --reps 10 --flow 0.2 2>&1 | tail -1
done
