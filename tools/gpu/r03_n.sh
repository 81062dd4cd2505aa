#!/bin/bash
# call N: fragment copy reads issued a GEMM group ahead (defer) vs in place
cd $GRAFT_REPO_ROOT
for v in _base _nodefer ""; do
  echo "== lib$v"
  NSFF_LIB=$GRAFT_REPO_ROOT/nsff_pl_amd/libnsff_hip$v.so timeout 300 python tools/debug/bwd_bench.py 131072 20 2>&1 | grep -E "static=|Error|error"
done
timeout 900 python -m pytest tests/test_field_grad.py -m gpu -x -q 2>&1 | tail -4
