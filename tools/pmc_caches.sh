#!/bin/bash
set -e
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
run() {
  name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/r2kr_c_$name -o p -- python3 tools/time_modes.py "kin+rect" > $OUT/r2kr_c_$name.log 2>&1
  echo "== $name: $*"
  python3 tools/pmc_summary.py $(find $OUT/r2kr_c_$name -name "p_counter_collection.csv" | head -1)
}
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum
run sqw SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT
