#!/bin/bash
# first run of the hand-scheduled kernel: parity subset with the 128-point tiles forced, then kernel timing A/B
TAG=${1:-b}
mkdir -p gpurun_out/r04_$TAG
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "f16x3-130" > gpurun_out/r04_$TAG/pytest130.log 2>&1; echo "pytest130 rc $?"
tail -15 gpurun_out/r04_$TAG/pytest130.log
timeout 300 python tools/bench_field.py --tile-points 130 > gpurun_out/r04_$TAG/field130.log 2>&1; cat gpurun_out/r04_$TAG/field130.log
timeout 300 python tools/bench_field.py --tile-points 131 > gpurun_out/r04_$TAG/field131.log 2>&1; cat gpurun_out/r04_$TAG/field131.log
