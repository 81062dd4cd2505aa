#!/bin/bash
# round-6 evidence (ONE gpurun call): un-profiled bench lines, rocprofv3 stats + counter passes of the default bench, the training
# step, the README configuration, the evaluation frame and the interpolation kernels (profiles/collect_r06.sh), their summary.
TAG=${1:-p}; shift || true
WHAT=${*:-stats pmc train readme readme_train interp eval}
mkdir -p gpurun_out/r06_$TAG
rocm-smi --showpower --showmaxpower --showclocks --showperflevel > gpurun_out/r06_$TAG/rocm_smi_idle.txt 2>&1
# the command the stats pass below profiles, un-profiled on the same box (the pair the roofline's avg_launch_ms is checked against)
python bench.py --no-cpu-baseline --no-aux > gpurun_out/r06_$TAG/bench_same_command.json 2>/dev/null
python bench.py --steps 400 --warmup 3 --no-cpu-baseline --no-aux > gpurun_out/r06_$TAG/bench_long.json 2>/dev/null
python bench.py --workload train --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r06_$TAG/bench_train_eager.json 2>/dev/null
python bench.py --workload train --graph --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r06_$TAG/bench_train_graph.json 2>/dev/null
tail -c 400 gpurun_out/r06_$TAG/bench_same_command.json; echo
bash profiles/collect_r06.sh $TAG $WHAT > gpurun_out/r06_$TAG/collect.log 2>&1
python profiles/summarize_r06.py gpurun_out/r06_$TAG > gpurun_out/r06_$TAG/pmc_summary.txt 2>&1
head -70 gpurun_out/r06_$TAG/pmc_summary.txt
# what travels back must stay small (64 MiB limit): keep the summaries and the stats tables, drop the raw traces / counter dumps
for d in bench train readme readme_train interp eval; do
  f=$(find gpurun_out/r06_$TAG/${d}_stats -name "*kernel_stats.csv" 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f gpurun_out/r06_$TAG/${d}_kernel_stats.csv
done
rm -rf gpurun_out/r06_$TAG/*_pmc gpurun_out/r06_$TAG/*_stats
ls gpurun_out/r06_$TAG
