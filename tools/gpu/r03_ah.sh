#!/bin/bash
# call AH: staggered GEMM with and without the weight refills
cd $GRAFT_REPO_ROOT
for v in "" _st _wnow _stnow _wnone; do
  echo "== lib$v"
  NSFF_LIB=$GRAFT_REPO_ROOT/nsff_pl_amd/libnsff_hip$v.so timeout 300 python tools/bench_field.py 2>&1 | tail -1
done
