#!/bin/bash
# hand-scheduled kernel: forced-tile parity subset, per-phase stamps (dynamic trunk), kernel timing A/B against the eight-wave form
TAG=${1:-c}
mkdir -p gpurun_out/r04_$TAG
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "f16x3-130" > gpurun_out/r04_$TAG/pytest130.log 2>&1; echo "pytest130 rc $?"; tail -3 gpurun_out/r04_$TAG/pytest130.log
NSFF_LIB=nsff_pl_amd/libnsff_hip_timing.so timeout 300 python tools/debug/h3a_timing.py dynamic > gpurun_out/r04_$TAG/timing_dynamic.txt 2>&1; cat gpurun_out/r04_$TAG/timing_dynamic.txt
timeout 300 python tools/bench_field.py --tile-points 130 > gpurun_out/r04_$TAG/field130.log 2>&1; cat gpurun_out/r04_$TAG/field130.log
timeout 300 python tools/bench_field.py --tile-points 131 > gpurun_out/r04_$TAG/field131.log 2>&1; tail -1 gpurun_out/r04_$TAG/field131.log
