#!/bin/bash
# call AE: f16x3 GEMM with staggered product order / LDS reads (no read in front of its MFMA, no extra registers)
cd $GRAFT_REPO_ROOT
for v in _nostag "" _nostag ""; do
  echo "== lib$v"
  NSFF_LIB=$GRAFT_REPO_ROOT/nsff_pl_amd/libnsff_hip$v.so timeout 300 python tools/bench_field.py 2>&1 | tail -1
  NSFF_LIB=$GRAFT_REPO_ROOT/nsff_pl_amd/libnsff_hip$v.so NSFF_TILE_POINTS=64 timeout 300 python tools/bench_field.py 2>&1 | tail -1
done
NSFF_LIB=$GRAFT_REPO_ROOT/nsff_pl_amd/libnsff_hip_timing.so timeout 300 python tools/debug/h3_timing.py 0 f16x3 2>&1 | tail -11 | cut -c1-230
timeout 300 python tools/debug/bwd_bench.py 131072 20 2>&1 | grep static
NSFF_LIB=$GRAFT_REPO_ROOT/nsff_pl_amd/libnsff_hip_nostag.so timeout 300 python tools/debug/bwd_bench.py 131072 20 2>&1 | grep static
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gradients.py tests/test_field_grad.py -m gpu -x -q 2>&1 | tail -2
