#!/bin/bash
# round 3, call G: weight ring carried across layer boundaries + head tile prefetched / three accumulator chains; A/B vs base
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_g; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_fast_mode.py tests/test_field_grad.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest parity+fast+field_grad rc=$?" >> $O/summary.txt
tail -5 $O/pytest.log >> $O/summary.txt
for rep in 1 2; do
for lib in "" nsff_pl_amd/libnsff_hip_base.so; do
  for tile in 0 64; do
    echo "== bench_field lib=${lib:-main} tile=$tile" >> $O/summary.txt
    NSFF_LIB=$lib timeout 300 python tools/bench_field.py --tile-points $tile --iters 20 2>&1 | grep -v amdgpu.ids >> $O/summary.txt
  done
done
done
for lib in "" nsff_pl_amd/libnsff_hip_base.so; do
  echo "== bench_field FAST lib=${lib:-main}" >> $O/summary.txt
  NSFF_LIB=$lib timeout 300 python tools/bench_field.py --precision f16 --iters 20 2>&1 | grep -v amdgpu.ids >> $O/summary.txt
  echo "== bwd_bench lib=${lib:-main}" >> $O/summary.txt
  NSFF_LIB=$lib timeout 300 python tools/debug/bwd_bench.py 131072 20 2>&1 | grep -v amdgpu.ids >> $O/summary.txt
done
echo "== h3_timing 0" >> $O/summary.txt
NSFF_LIB=nsff_pl_amd/libnsff_hip_timing.so timeout 300 python tools/debug/h3_timing.py 0 f16x3 2>&1 | grep -v amdgpu.ids >> $O/summary.txt
cat $O/summary.txt
