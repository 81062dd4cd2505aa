#!/bin/bash
# round 3, call D: the ping-pong field kernel (tile_points = 131): parity, gradients through its training forward, timing A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "131" > $O/pytest_pp.log 2>&1; echo "pytest parity 131 rc=$?" >> $O/summary.txt
tail -8 $O/pytest_pp.log >> $O/summary.txt
NSFF_TILE_POINTS=131 timeout 900 python -m pytest tests/test_field_grad.py tests/test_gradients.py tests/test_losses.py -m gpu -x -q > $O/pytest_pp_grad.log 2>&1; echo "pytest grad (131 SAVE) rc=$?" >> $O/summary.txt
tail -8 $O/pytest_pp_grad.log >> $O/summary.txt
timeout 600 python -m pytest tests/test_field_grad.py tests/test_gradients.py tests/test_dist_gpu.py -m gpu -x -q > $O/pytest_grad.log 2>&1; echo "pytest grad+dist (default) rc=$?" >> $O/summary.txt
tail -4 $O/pytest_grad.log >> $O/summary.txt
for tile in 0 131 64 0 131; do
  echo "== bench_field tile=$tile" >> $O/summary.txt
  timeout 300 python tools/bench_field.py --tile-points $tile --iters 20 >> $O/summary.txt 2>&1
done
for tile in 0 131 64; do
  echo "== bwd_bench tile=$tile" >> $O/summary.txt
  NSFF_TILE_POINTS=$tile timeout 300 python tools/debug/bwd_bench.py 131072 20 >> $O/summary.txt 2>&1
done
for tile in 0 131 0 131; do
  echo "== bench.py render tile=$tile" >> $O/summary.txt
  timeout 300 python bench.py --tile-points $tile --steps 20 --warmup 5 --no-cpu-baseline --no-aux 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*' >> $O/summary.txt
  echo "== bench.py train tile=$tile" >> $O/summary.txt
  timeout 300 python bench.py --tile-points $tile --workload train --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> $O/summary.txt
done
cat $O/summary.txt
