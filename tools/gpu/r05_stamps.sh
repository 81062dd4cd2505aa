#!/bin/bash
# round 5: phase stamps of the timing build (make -C nsff_pl_amd/csrc timing), persistent launch and NSFF_NO_PERSIST=1.  usage: bash tools/gpu/r05_stamps.sh <tag>
TAG=${1:-s}; export NSFF_LIB=nsff_pl_amd/libnsff_hip_timing.so
for w in static dynamic_tb; do
  timeout 200 python tools/debug/h3a_timing.py $w 2>&1 | grep -v -e "^  phase [0-9 ]*[AB]16" -e amdgpu.ids > gpurun_out/r05_stamps_${TAG}_persist_$w.txt
  NSFF_NO_PERSIST=1 timeout 200 python tools/debug/h3a_timing.py $w 2>&1 | grep -v -e "^  phase [0-9 ]*[AB]16" -e amdgpu.ids > gpurun_out/r05_stamps_${TAG}_tile_$w.txt
done
tail -n 3 gpurun_out/r05_stamps_${TAG}_persist_dynamic_tb.txt gpurun_out/r05_stamps_${TAG}_tile_dynamic_tb.txt gpurun_out/r05_stamps_${TAG}_persist_static.txt
