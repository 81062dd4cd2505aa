#!/bin/bash
# Memory-path counters (TA/TCP/TD/TCC) for a field kernel; one pass per block.
set -u
TAG=${1:-mem}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD=${PMC_CMD:-"python $ROOT/tools/bench_field.py --iters 2"}
run() { local name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT -o $name -- $CMD > $OUT/$name.log 2>&1; }
run ta   TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum GRBM_GUI_ACTIVE
run tcp  TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum
run tcp2 TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run td   TD_TD_BUSY_sum TD_TC_STALL_sum TD_SPI_STALL_sum
run tcc  TCC_BUSY_sum TCC_TAG_STALL_sum TCC_REQ_sum
run sq   SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_IFETCH_LEVEL SQC_ICACHE_MISSES SQC_ICACHE_REQ
ls $OUT | head -3
