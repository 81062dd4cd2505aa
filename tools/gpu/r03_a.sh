#!/bin/bash
# round 3, call A: correctness of the interleaved activation / gradient saves + the 128-point training forward, then A/B timing
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03_a
O=gpurun_out/r03_a
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_field_grad.py tests/test_gradients.py tests/test_losses.py -m gpu -x -q > $O/pytest_grad.log 2>&1; echo "pytest grad rc=$?" >> $O/summary.txt
tail -3 $O/pytest_grad.log >> $O/summary.txt
NSFF_TILE_POINTS=64 timeout 600 python -m pytest tests/test_field_grad.py tests/test_gradients.py -m gpu -x -q > $O/pytest_grad64.log 2>&1; echo "pytest grad (64-pt SAVE) rc=$?" >> $O/summary.txt
tail -3 $O/pytest_grad64.log >> $O/summary.txt
for lib in "" nsff_pl_amd/libnsff_hip_nointer.so; do
  for tile in 0 64; do
    echo "== lib=${lib:-main} tile=$tile" >> $O/summary.txt
    NSFF_LIB=$lib NSFF_TILE_POINTS=$tile timeout 300 python tools/debug/bwd_bench.py 131072 20 >> $O/summary.txt 2>&1
  done
done
for lib in "" nsff_pl_amd/libnsff_hip_nointer.so; do
  for tile in 0 64; do
    echo "== train step lib=${lib:-main} tile=$tile" >> $O/summary.txt
    NSFF_LIB=$lib NSFF_TILE_POINTS=$tile timeout 300 python bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ms_per_step', d['ms_per_step'])" >> $O/summary.txt
  done
done
cat $O/summary.txt
