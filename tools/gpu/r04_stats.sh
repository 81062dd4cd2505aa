#!/bin/bash
# rocprofv3 kernel stats of a short headline run (per-kernel average durations), printed; raw traces deleted
TAG=${1:-s}; D=gpurun_out/r04_$TAG; mkdir -p $D; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$D/raw -o st -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-aux --no-cpu-baseline > $GRAFT_REPO_ROOT/$D/bench.json 2> $GRAFT_REPO_ROOT/$D/err.log
cd $GRAFT_REPO_ROOT
f=$(find $D/raw -name '*kernel_stats.csv' | head -1); cp "$f" $D/kernel_stats.csv; rm -rf $D/raw
python - <<P
import csv
rows=list(csv.DictReader(open("$D/kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:26]:
    print("%-90s calls %5s avg %9.1f us  %5.1f%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
P
tail -c 300 $D/bench.json
