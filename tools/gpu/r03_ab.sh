#!/bin/bash
# call AB: second half of the head weights requested behind the last GEMM (ring2) vs head biases only
cd $GRAFT_REPO_ROOT
for v in _base _nor2 "" _base _nor2 ""; do
  echo "== lib$v"
  NSFF_LIB=$GRAFT_REPO_ROOT/nsff_pl_amd/libnsff_hip$v.so timeout 300 python tools/bench_field.py 2>&1 | tail -1
done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
