#!/bin/bash
# Hardware counters of the training kernels (training forward, nsff_field_bwd_kernel, nsff_wgrad_kernel):
# three rocprofv3 passes (--kernel-trace + --pmc only), eager training steps.
# Usage: bash profiles/collect_pmc_train.sh <tag>  -> gpurun_out/pmc_<tag>/
set -u
TAG=${1:-r01d_train}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --workload train --steps 2 --warmup 1 --no-cpu-baseline"
run() { local name=$1; shift
  timeout 250 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT -o $name -- $CMD > $OUT/$name.log 2>&1
}
run mfma  SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
