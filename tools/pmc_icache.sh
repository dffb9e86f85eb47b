#!/bin/bash
# Instruction-cache counters of one workload of tools/time_modes.py (development aid).
#   B=16384 bash tools/pmc_icache.sh "<workload name filter>"
set -e
W=$1
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -s KILL 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $OUT/r2_ic -o p -- python3 tools/time_modes.py "$W" > $OUT/r2_ic.log 2>&1
python3 tools/pmc_summary.py $(find $OUT/r2_ic -name "p_counter_collection.csv" | head -1)
