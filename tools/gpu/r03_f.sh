#!/bin/bash
# round 3, call F: ping-pong kernel, weight ring 8 vs 4, phase timing
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_f; mkdir -p $O
for lib in "" nsff_pl_amd/libnsff_hip_ring4.so; do
  for tile in 131 0; do
    echo "== bench_field lib=${lib:-main(ring8)} tile=$tile" >> $O/summary.txt
    NSFF_LIB=$lib timeout 300 python tools/bench_field.py --tile-points $tile --iters 20 2>&1 | grep -v amdgpu.ids >> $O/summary.txt
  done
done
echo "== h3_timing 131" >> $O/summary.txt
NSFF_LIB=nsff_pl_amd/libnsff_hip_timing.so timeout 300 python tools/debug/h3_timing.py 131 f16x3 2>&1 | grep -v amdgpu.ids >> $O/summary.txt
echo "== h3_timing 0" >> $O/summary.txt
NSFF_LIB=nsff_pl_amd/libnsff_hip_timing.so timeout 300 python tools/debug/h3_timing.py 0 f16x3 2>&1 | grep -v amdgpu.ids >> $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "131" > $O/pytest_pp.log 2>&1; echo "pytest parity 131 rc=$?" >> $O/summary.txt
timeout 900 python -m pytest tests/test_optim.py tests/test_field_grad.py tests/test_dist_gpu.py -m gpu -x -q > $O/pytest_misc.log 2>&1; echo "pytest optim+field_grad+dist rc=$?" >> $O/summary.txt
tail -6 $O/pytest_misc.log >> $O/summary.txt
cat $O/summary.txt
