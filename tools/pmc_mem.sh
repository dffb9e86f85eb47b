#!/bin/bash
# Development aid: vector-memory path counters of the pipelined bench (one run per counter set).
# Usage on the box:  bash tools/pmc_mem.sh <tag> [bench args...]
set -e
TAG=$1; shift
OUT=gpurun_out/r2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
run() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/${TAG}_$name -o p -- python3 bench.py --no-cpu "$@" > $OUT/${TAG}_$name.log 2>&1
  echo "== $name: $CTRS"
  python3 tools/pmc_summary.py $(find $OUT/${TAG}_$name -name "p_counter_collection.csv" | head -1)
}
CTRS="TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" run ta "$@"
CTRS="SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_CYCLES_SMEM" run sqv "$@"
