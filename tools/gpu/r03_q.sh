#!/bin/bash
# call Q: what is left between the training nodes (kernel stats of the eager step)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_q -o train -- python $R/bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/r03_q.log 2>&1
cd $R
python - <<'PY'
import csv, glob
for p in glob.glob('gpurun_out/r03_q/*kernel_stats.csv'):
    rows = list(csv.DictReader(open(p)))
    rows.sort(key=lambda r: -float(r['TotalDurationNs']))
    steps = 13
    tot=sum(float(r['TotalDurationNs']) for r in rows)
    print('total ms/step', tot/steps/1e6, 'launches/step', sum(int(r['Calls']) for r in rows)/steps)
    for r in rows[7:70]:
        print(f"{r['Name'][:150]:150s} calls/step {int(r['Calls'])/steps:6.1f} us/step {float(r['TotalDurationNs'])/steps/1e3:7.1f}")
PY
