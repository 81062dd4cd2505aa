#!/bin/bash
TAG=${1:-t}; shift; mkdir -p gpurun_out/r04_$TAG
timeout 900 python -m pytest "$@" -x -q > gpurun_out/r04_$TAG/pytest.log 2>&1; echo "pytest rc $?"; tail -25 gpurun_out/r04_$TAG/pytest.log
