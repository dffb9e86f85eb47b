#!/bin/bash
# Counter passes of one workload of tools/time_modes.py (development aid).  Usage on the box:
#   B=16384 bash tools/pmc_workload.sh <tag> "<workload name filter>"
set -e
TAG=$1; W=$2
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
run() {
  name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/${TAG}_w_$name -o p -- python3 tools/time_modes.py "$W" > $OUT/${TAG}_w_$name.log 2>&1
  echo "== $name: $*"
  python3 tools/pmc_summary.py $(find $OUT/${TAG}_w_$name -name "p_counter_collection.csv" | head -1)
}
run stall SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_IFETCH
run mix SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_SMEM
