#!/bin/bash
# call AI: four waves of 64 neurons x 128 points (one wave per SIMD, half the LDS operand traffic) with staggered products and
# eight k-steps of weights in flight, against the default eight-wave tiling
cd $GRAFT_REPO_ROOT
for t in 0 128 0 128; do
  echo "== tile $t"
  timeout 300 python tools/bench_field.py --tile-points $t 2>&1 | tail -1
done
echo "== tile 128, plain GEMM loop (4-deep ring, reads in front of their MFMAs)"
NSFF_LIB=$GRAFT_REPO_ROOT/nsff_pl_amd/libnsff_hip_wide4.so timeout 300 python tools/bench_field.py --tile-points 128 2>&1 | tail -1
timeout 300 python tools/debug/h3_parity_tile.py 128 2>&1 | tail -3
