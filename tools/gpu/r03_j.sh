#!/bin/bash
# round 3, call J: binned far path of the splat
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_j; mkdir -p $O
timeout 1500 python -m pytest tests/test_interpolate.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest interpolate rc=$?" >> $O/summary.txt
grep -E "^FAILED|^ERROR|passed|failed" $O/pytest.log | tail -10 >> $O/summary.txt
grep -E "^E  " $O/pytest.log | head -20 >> $O/summary.txt
for bin in 1 0; do
  for flow in 0.02 0.2; do
    for planes in 192 256; do
      echo "== bench_interp binning=$bin flow=$flow planes=$planes" >> $O/summary.txt
      NSFF_SPLAT_BINNING=$bin timeout 300 python tools/bench_interp.py --flow $flow --planes $planes --reps 5 2>&1 | grep -v amdgpu.ids >> $O/summary.txt
    done
  done
done
cat $O/summary.txt
