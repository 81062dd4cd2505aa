#!/bin/bash
# Hardware-counter passes for the field kernel (run on the GPU box through gpurun).
# One rocprofv3 run per counter group, --kernel-trace only (no sys/hip tracing with --pmc).
# Usage: bash profiles/collect_pmc.sh <tag>   -> gpurun_out/pmc_<tag>/<group>_counter_collection.csv
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD=${PMC_CMD:-"python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-aux"}
run() { # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT -o $name -- $CMD > $OUT/$name.log 2>&1
}
run mfma  SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU GRBM_GUI_ACTIVE
run wait  SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
run lds   SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc   TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
ls -la $OUT
