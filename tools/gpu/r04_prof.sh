#!/bin/bash
# round-4 evidence: parity of the forced tilings, power / clock readout, rocprofv3 stats + counter passes of the default bench
TAG=${1:-p}
mkdir -p gpurun_out/r04_$TAG
rocm-smi --showpower --showmaxpower --showclocks --showperflevel > gpurun_out/r04_$TAG/rocm_smi_idle.txt 2>&1
# the command the stats pass below profiles, un-profiled on the same box (the pair the roofline's avg_launch_ms is checked against)
python bench.py --no-cpu-baseline --no-aux > gpurun_out/r04_$TAG/bench_same_command.json 2>/dev/null
python bench.py --steps 400 --warmup 3 --no-cpu-baseline --no-aux > gpurun_out/r04_$TAG/bench_long.json 2>/dev/null
tail -c 400 gpurun_out/r04_$TAG/bench_same_command.json; echo
bash profiles/collect_r04.sh $TAG stats pmc > gpurun_out/r04_$TAG/collect.log 2>&1
python profiles/summarize_r03.py gpurun_out/r04_$TAG > gpurun_out/r04_$TAG/pmc_summary.txt 2>&1
head -60 gpurun_out/r04_$TAG/pmc_summary.txt
# what travels back must stay small (64 MiB limit): keep the summaries and the stats tables, drop the raw traces / counter dumps
cp gpurun_out/r04_$TAG/bench_stats/bench_kernel_stats.csv gpurun_out/r04_$TAG/bench_kernel_stats.csv
rm -rf gpurun_out/r04_$TAG/bench_pmc gpurun_out/r04_$TAG/fast_pmc gpurun_out/r04_$TAG/fast_stats gpurun_out/r04_$TAG/bench_stats
