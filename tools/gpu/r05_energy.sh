#!/bin/bash
# round 5: time and clock of body variants that do LESS (results garbage; tools/h3asm/gen.py H3A_EXP=..., make variant NAME=...):
# how the power-limited part answers fewer MFMAs / no epilogue stores / no epilogue at all.  usage: bash tools/gpu/r05_energy.sh v1 v2 ...
mkdir -p gpurun_out/r05_energy
for i in 1 2; do
for v in "$@"; do
  lib=nsff_pl_amd/libnsff_hip_$v.so; [ "$v" = base ] && lib=nsff_pl_amd/libnsff_hip.so
  echo "== $v (round $i)"; NSFF_LIB=$lib timeout 300 python tools/bench_field.py --tile-points 130 --iters 300 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r05_energy/$v.log
done
done
