#!/bin/bash
# round 3, call E: micro-interleaved ping-pong kernel; graph determinism; CPU-side fixes (egress pool, optimizer state, pending)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "131" > $O/pytest_pp.log 2>&1; echo "pytest parity 131 rc=$?" >> $O/summary.txt
tail -4 $O/pytest_pp.log >> $O/summary.txt
for tile in 0 131 0 131; do
  echo "== bench_field tile=$tile" >> $O/summary.txt
  timeout 300 python tools/bench_field.py --tile-points $tile --iters 20 2>&1 | grep -v amdgpu.ids >> $O/summary.txt
done
NSFF_TILE_POINTS=131 timeout 900 python -m pytest tests/test_field_grad.py tests/test_gradients.py -m gpu -x -q > $O/pytest_pp_grad.log 2>&1; echo "pytest grad (131 SAVE) rc=$?" >> $O/summary.txt
tail -4 $O/pytest_pp_grad.log >> $O/summary.txt
for tile in 0 131; do
  echo "== bwd_bench tile=$tile" >> $O/summary.txt
  NSFF_TILE_POINTS=$tile timeout 300 python tools/debug/bwd_bench.py 131072 20 2>&1 | grep -v amdgpu.ids >> $O/summary.txt
done
echo "== graph determinism" >> $O/summary.txt
timeout 300 python tools/debug/graph_determinism.py 2>&1 | grep -v amdgpu.ids >> $O/summary.txt
timeout 900 python -m pytest tests/test_optim.py tests/test_field_grad.py tests/test_dist_gpu.py -m gpu -x -q > $O/pytest_misc.log 2>&1; echo "pytest optim+field_grad+dist rc=$?" >> $O/summary.txt
tail -6 $O/pytest_misc.log >> $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "egress or free_running" > $O/pytest_par2.log 2>&1; echo "pytest egress+free-running rc=$?" >> $O/summary.txt
tail -6 $O/pytest_par2.log >> $O/summary.txt
cat $O/summary.txt
