#!/bin/bash
# timing experiments: variants of the hand-scheduled body (results are garbage, only time is read)
TAG=${1:-e}; shift
mkdir -p gpurun_out/r04_$TAG
for v in "$@"; do
  lib=nsff_pl_amd/libnsff_hip_$v.so; [ "$v" = base ] && lib=nsff_pl_amd/libnsff_hip.so
  echo "== $v"; NSFF_LIB=$lib timeout 200 python tools/bench_field.py --tile-points 130 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_$TAG/$v.log
done
