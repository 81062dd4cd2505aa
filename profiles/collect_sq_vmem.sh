#!/bin/bash
# VMEM issue-side counters (SQ only; the TA/TCP "_sum" counters hang rocprofv3 on this box).
set -u
TAG=${1:-sqv}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD=${PMC_CMD:-"python $ROOT/tools/bench_field.py --iters 2"}
timeout 120 rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT -o sq -- $CMD > $OUT/sq.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_IFETCH_LEVEL SQ_IFETCH SQC_ICACHE_MISSES SQC_ICACHE_REQ SQC_ICACHE_HITS SQ_WAIT_INST_LDS SQ_INSTS_LDS --output-format csv -d $OUT -o ic -- $CMD > $OUT/ic.log 2>&1
