#!/bin/bash
# call V: XCD-aware (tile, plane group) order of the splat workgroups; parity of the interpolate suite
cd $GRAFT_REPO_ROOT
for v in _x0 "" _x1; do
  echo "== lib$v"
  NSFF_LIB=$GRAFT_REPO_ROOT/nsff_pl_amd/libnsff_hip$v.so timeout 300 python tools/bench_interp.py --planes 256 --reps 10 2>&1 | tail -1
  NSFF_LIB=$GRAFT_REPO_ROOT/nsff_pl_amd/libnsff_hip$v.so timeout 300 python tools/bench_interp.py --planes 256 --reps 10 --flow 0.2 2>&1 | tail -1
done
timeout 900 python -m pytest tests/test_interpolate.py -m gpu -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03_v -o fetch -- python $GRAFT_REPO_ROOT/tools/bench_interp.py --reps 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, collections
c=collections.defaultdict(lambda:[0,0])
for r in csv.DictReader(open('gpurun_out/r03_v/fetch_counter_collection.csv')):
    if 'splat' in r['Kernel_Name'] or 'mpi' in r['Kernel_Name']:
        k=r['Kernel_Name'].split('(')[0][-30:]; c[k][0]+=float(r['Counter_Value']); c[k][1]+=1
for k,(v,n) in c.items(): print(k, 'FETCH_SIZE per launch MB (x2 rule)', 2*v/n/1024)
PY
