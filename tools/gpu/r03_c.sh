#!/bin/bash
# round 3, call C: new backward epilogue + packed fragment copy + in-kernel visibility + nccl world-1 tests; timing
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_c; mkdir -p $O
timeout 900 python -m pytest tests/test_field_grad.py tests/test_gradients.py tests/test_losses.py tests/test_dist_gpu.py -m gpu -x -q > $O/pytest_a.log 2>&1; echo "pytest grad+dist rc=$?" >> $O/summary.txt
tail -12 $O/pytest_a.log >> $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "g5 or vis or golden" > $O/pytest_b.log 2>&1; echo "pytest parity(g5) rc=$?" >> $O/summary.txt
tail -5 $O/pytest_b.log >> $O/summary.txt
for lib in "" nsff_pl_amd/libnsff_hip_nointer.so; do
    echo "== lib=${lib:-main}" >> $O/summary.txt
    NSFF_LIB=$lib timeout 300 python tools/debug/bwd_bench.py 131072 20 >> $O/summary.txt 2>&1
done
echo "== bwd_timing" >> $O/summary.txt
NSFF_LIB=nsff_pl_amd/libnsff_hip_timing.so timeout 300 python tools/debug/bwd_timing.py >> $O/summary.txt 2>&1
echo "== h3_timing save" >> $O/summary.txt
NSFF_LIB=nsff_pl_amd/libnsff_hip_timing.so timeout 300 python tools/debug/h3_timing.py 0 f16x3 save >> $O/summary.txt 2>&1
echo "== train step" >> $O/summary.txt
timeout 300 python bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> $O/summary.txt
cat $O/summary.txt
