#!/bin/bash
# call K: interpolate parity after the crafted-case fix; training-step kernel stats + counters
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_interpolate.py -m gpu -x -q 2>&1 | tail -5
bash profiles/collect_r03.sh a train > gpurun_out/collect_train.log 2>&1
python profiles/summarize_r03.py gpurun_out/r03_a > gpurun_out/r03_a_summary.txt 2>&1
tail -3 gpurun_out/r03_a/train_stats.log
python - <<'PY'
import csv, glob
for p in glob.glob('gpurun_out/r03_a/train_stats/*kernel_stats.csv'):
    rows = list(csv.DictReader(open(p)))
    rows.sort(key=lambda r: -float(r['TotalDurationNs']))
    for r in rows[:22]:
        print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} total_ms {float(r['TotalDurationNs'])/1e6:8.2f} avg_us {float(r['AverageNs'])/1e3:8.1f} {r['Percentage']}%")
PY
