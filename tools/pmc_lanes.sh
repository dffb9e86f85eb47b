#!/bin/bash
# Development aid: instruction and stall counters of the bench kernel at several lane-group widths (one run each).
# Usage on the box:  bash tools/pmc_lanes.sh <tag> <batch> <lanes...>
set -e
TAG=$1; shift
B=$1; shift
OUT=gpurun_out/r2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for L in "$@"; do
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d $OUT/${TAG}_l$L -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu --pipeline 1 --lanes $L --batch $B > $OUT/${TAG}_l$L.log 2>&1
  echo "== lanes $L batch $B (sum over 3 identical launches)"
  python3 tools/pmc_summary.py $(find $OUT/${TAG}_l$L -name "p_counter_collection.csv" | head -1)
  grep -o '"lone_launch_ms": [0-9.]*' $OUT/${TAG}_l$L.log
done
