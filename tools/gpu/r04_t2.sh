#!/bin/bash
# pytest selection, then the headline bench leg (no aux, no CPU baseline) three times
TAG=${1:-t}; shift; mkdir -p gpurun_out/r04_$TAG
timeout 1500 python -m pytest "$@" -x -q > gpurun_out/r04_$TAG/pytest.log 2>&1; echo "pytest rc $?"; tail -12 gpurun_out/r04_$TAG/pytest.log
for i in 1 2 3; do python bench.py --steps 200 --warmup 20 --no-aux --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_$TAG/bench_$i.json
python - <<P
import json; d=json.load(open("gpurun_out/r04_$TAG/bench_$i.json")); r=d["roofline"]
print("ms/step %.4f  launch %.4f ms  clk %.3f  frac %.4f  %s" % (d["ms_per_step"], r["avg_launch_ms"], r.get("clock_ghz") or 0, r["frac"], r.get("kernel")))
P
done
