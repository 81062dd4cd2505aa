#!/bin/bash
# call M: transposing-read fragment copies (training forward + field backward): same-box A/B, then the gradient suites
cd $GRAFT_REPO_ROOT
for v in _base ""; do
  echo "== lib$v"
  NSFF_LIB=$GRAFT_REPO_ROOT/nsff_pl_amd/libnsff_hip$v.so timeout 300 python tools/debug/bwd_bench.py 131072 20 2>&1 | grep -E "static=|Error|error"
done
timeout 900 python -m pytest tests/test_field_grad.py tests/test_gradients.py -m gpu -x -q 2>&1 | tail -8
