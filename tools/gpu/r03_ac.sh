#!/bin/bash
# call AC: B-operand (activation) LDS reads pinned one k-step ahead of their MFMAs
cd $GRAFT_REPO_ROOT
for v in _nox "" _nox ""; do
  echo "== lib$v"
  NSFF_LIB=$GRAFT_REPO_ROOT/nsff_pl_amd/libnsff_hip$v.so timeout 300 python tools/bench_field.py 2>&1 | tail -1
  NSFF_LIB=$GRAFT_REPO_ROOT/nsff_pl_amd/libnsff_hip$v.so NSFF_TILE_POINTS=64 timeout 300 python tools/bench_field.py 2>&1 | tail -1
  NSFF_LIB=$GRAFT_REPO_ROOT/nsff_pl_amd/libnsff_hip$v.so timeout 300 python tools/bench_field.py --precision f16 2>&1 | tail -1
done
timeout 300 python tools/debug/bwd_bench.py 131072 20 2>&1 | grep static
NSFF_LIB=$GRAFT_REPO_ROOT/nsff_pl_amd/libnsff_hip_nox.so timeout 300 python tools/debug/bwd_bench.py 131072 20 2>&1 | grep static
