#!/bin/bash
# call AG: what-if timings of the forward kernel (results are garbage): no weight refills / no B-fragment reads / neither
cd $GRAFT_REPO_ROOT
for v in "" _wnow _wnox _wnone; do
  echo "== lib$v"
  NSFF_LIB=$GRAFT_REPO_ROOT/nsff_pl_amd/libnsff_hip$v.so timeout 300 python tools/bench_field.py 2>&1 | tail -1
done
