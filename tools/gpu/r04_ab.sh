#!/bin/bash
# A/B of environment switches on one box: bench.py (headline leg only) plain and with each `VAR=VALUE` given, interleaved
mkdir -p gpurun_out/r04_ab; rm -f gpurun_out/r04_ab/*
for i in 1 2 3; do
  python bench.py --steps 200 --warmup 20 --no-aux --no-cpu-baseline > gpurun_out/r04_ab/base_$i.json 2>gpurun_out/r04_ab/base_$i.err
  for kv in "$@"; do
    name=$(echo "$kv" | tr '/' '-')
    env $kv python bench.py --steps 200 --warmup 20 --no-aux --no-cpu-baseline > gpurun_out/r04_ab/${name}_$i.json 2>gpurun_out/r04_ab/${name}_$i.err
  done
done
python - <<'P'
import json,glob,os
rows={}
for f in sorted(glob.glob("gpurun_out/r04_ab/*.json")):
    k=os.path.basename(f).rsplit("_",1)[0]
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        rows.setdefault(k,[]).append((d["ms_per_step"], r["avg_launch_ms"], r.get("clock_ghz") or 0))
    except Exception as e: print(k, f, "ERR", e)
for k,v in rows.items():
    n=len(v); print("%-28s ms/step %.4f  launch %.4f ms  clk %.3f  (n=%d)" % (k, sum(x[0] for x in v)/n, sum(x[1] for x in v)/n, sum(x[2] for x in v)/n, n))
P
