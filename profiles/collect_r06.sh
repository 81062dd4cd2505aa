#!/bin/bash
# Round-6 evidence collection (run on the GPU box through gpurun):   bash profiles/collect_r06.sh <tag> [what...]
#   stats   rocprofv3 --kernel-trace --stats of the default bench  -> gpurun_out/r06_<tag>/bench_*
#   pmc     counter passes (one rocprofv3 run per group, --kernel-trace only) of the default bench (f16x3 field
#           kernel + the per-ray kernels)
#   readme  kernel stats + FETCH / WRITE of the reference's README configuration (512x288, 128 samples, view directions)
#   interp  FETCH / WRITE passes of the time-interpolation kernels (tools/bench_interp.py)
#   eval    kernel stats + FETCH / WRITE of one 512x288 test-time frame with the visibility branch on (C3)
#   train   kernel stats + FETCH / WRITE / mfma passes of the training step
# Summaries: python profiles/summarize_r06.py gpurun_out/r06_<tag>  > profiles/r06_<tag>_summary.txt
set -u
TAG=${1:-a}; shift || true
WHAT=${*:-stats pmc interp train readme eval readme_train}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-aux"
pmc() { # dir, name, cmd, counters...
  local dir=$1 name=$2 cmd=$3; shift 3
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$dir -o $name -- $cmd > $OUT/$dir-$name.log 2>&1
}
groups() { # dir, cmd
  pmc $1 mfma  "$2" SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU GRBM_GUI_ACTIVE
  pmc $1 lds   "$2" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
  pmc $1 fetch "$2" FETCH_SIZE
  pmc $1 write "$2" WRITE_SIZE
  pmc $1 tcc   "$2" TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
}
for w in $WHAT; do
  case $w in
    stats)
      rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -o bench -- python $ROOT/bench.py --no-cpu-baseline --no-aux > $OUT/bench_stats.log 2>&1
      ;;
    pmc)
      groups bench_pmc "$BENCH" ;;
    readme)
      rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/readme_stats -o readme -- python $ROOT/tools/debug/readme_frame_timing.py 2 > $OUT/readme_stats.log 2>&1
      pmc readme_pmc fetch "python $ROOT/tools/debug/readme_frame_timing.py 1" FETCH_SIZE
      pmc readme_pmc write "python $ROOT/tools/debug/readme_frame_timing.py 1" WRITE_SIZE ;;
    interp)
      rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/interp_stats -o interp -- python $ROOT/tools/bench_interp.py --reps 2 > $OUT/interp_stats.log 2>&1
      pmc interp_pmc fetch "python $ROOT/tools/bench_interp.py --reps 2" FETCH_SIZE
      pmc interp_pmc write "python $ROOT/tools/bench_interp.py --reps 2" WRITE_SIZE ;;
    eval)
      rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/eval_stats -o eval -- python $ROOT/bench.py --workload eval --steps 2 --warmup 1 --no-cpu-baseline --no-aux > $OUT/eval_stats.log 2>&1
      pmc eval_pmc fetch "python $ROOT/bench.py --workload eval --steps 1 --warmup 1 --no-cpu-baseline --no-aux" FETCH_SIZE
      pmc eval_pmc write "python $ROOT/bench.py --workload eval --steps 1 --warmup 1 --no-cpu-baseline --no-aux" WRITE_SIZE ;;
    readme_train)
      rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/readme_train_stats -o readme_train -- python $ROOT/tools/debug/readme_train_timing.py 10 > $OUT/readme_train_stats.log 2>&1 ;;
    train)
      rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_stats -o train -- python $ROOT/bench.py --workload train --graph --steps 10 --warmup 3 --no-cpu-baseline > $OUT/train_stats.log 2>&1
      T="python $ROOT/bench.py --workload train --steps 2 --warmup 2 --no-cpu-baseline"
      pmc train_pmc mfma "$T" SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
      pmc train_pmc fetch "$T" FETCH_SIZE
      pmc train_pmc write "$T" WRITE_SIZE ;;
  esac
done
ls $OUT
