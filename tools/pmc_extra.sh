#!/bin/bash
# Extra counter passes of one lone launch of the bench workload (development aid): issue stalls, instruction cache,
# executed fp64 instruction mix.  Each pass is its own run.  Usage on the box:  bash tools/pmc_extra.sh r01
set -e
TAG=${1:-r01}
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
run() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/${TAG}_x_$name -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu > $OUT/${TAG}_x_$name.log 2>&1
  echo "== $name: $*"
  python3 tools/pmc_summary.py $(find $OUT/${TAG}_x_$name -name "p_counter_collection.csv" | head -1)
}
run stall SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_IFETCH
run icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
run mix SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU
