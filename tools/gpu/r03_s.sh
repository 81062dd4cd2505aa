#!/bin/bash
# call S: time-code gather backward with ordered compaction
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_field_grad.py tests/test_gradients.py -m gpu -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_s -o train -- python $R/bench.py --workload train --graph --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/r03_s.log 2>&1
cd $R
python - <<'PY'
import csv, glob
for p in glob.glob('gpurun_out/r03_s/*kernel_stats.csv'):
    rows = list(csv.DictReader(open(p)))
    for r in rows:
        if 'time_rows' in r['Name'] or 'flow_grad' in r['Name'] or 'input_bwd' in r['Name']:
            print(f"{r['Name'][:100]:100s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
timeout 600 python bench.py --workload train --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
timeout 600 python bench.py --workload train --graph --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
