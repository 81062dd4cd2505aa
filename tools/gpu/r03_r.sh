#!/bin/bash
# call R: head weight-gradient kernel with eight steps of fragments in flight
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_field_grad.py tests/test_gradients.py -m gpu -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_r -o train -- python $R/bench.py --workload train --graph --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/r03_r.log 2>&1
tail -1 $R/gpurun_out/r03_r.log | cut -c1-330
cd $R
python - <<'PY'
import csv, glob
for p in glob.glob('gpurun_out/r03_r/*kernel_stats.csv'):
    rows = list(csv.DictReader(open(p)))
    rows.sort(key=lambda r: -float(r['TotalDurationNs']))
    for r in rows[:12]:
        print(f"{r['Name'][:100]:100s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
