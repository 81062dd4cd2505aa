#!/bin/bash
# call AA: head biases requested at head entry
cd $GRAFT_REPO_ROOT
for v in _base "" _base ""; do
  echo "== lib$v"
  NSFF_LIB=$GRAFT_REPO_ROOT/nsff_pl_amd/libnsff_hip$v.so timeout 300 python tools/bench_field.py 2>&1 | tail -1
  NSFF_LIB=$GRAFT_REPO_ROOT/nsff_pl_amd/libnsff_hip$v.so timeout 300 python tools/bench_field.py --precision f16 2>&1 | tail -1
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fast_mode.py -m gpu -x -q 2>&1 | tail -2
