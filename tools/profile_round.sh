#!/bin/bash
# Profiles of one round on the GPU box.  Usage (on the box, from the repo root):  bash tools/profile_round.sh r03 "2 3 4a 4b 5"
#   1. rocprofv3 --kernel-trace --stats of the default bench command (the number the driver records)
#   2. per configuration: counter passes over tools/pmc_run.py (3 identical launches of the configuration's batch through
#      alilqr_solve_batch; one run per counter set: FETCH_SIZE, WRITE_SIZE, wave-cycle shares, instruction mix)
#   3. the FETCH_SIZE / WRITE_SIZE calibration program (tools/calib/fetch_calib) under the same two counters
# Raw output goes to gpurun_out/<tag>_*; tools/profile_summarise.py <tag> turns it into the files under profiles/.
# Every GPU step has its own time limit and the script stops at the first failure.
set -e
TAG=${1:-r03}
CONFIGS=${2:-"2"}
OUT=gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_kt -o kt -- python3 bench.py > $OUT/${TAG}_kt.log 2>&1
echo "kernel trace done"
pass() {  # pass <config> <name> <counters...>
  cfg=$1; name=$2; shift 2
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/${TAG}_c${cfg}_$name -o p -- python3 tools/pmc_run.py --config $cfg > $OUT/${TAG}_c${cfg}_$name.log 2>&1
}
for c in $CONFIGS; do
  pass $c fetch FETCH_SIZE
  pass $c write WRITE_SIZE
  pass $c stall SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVES
  pass $c mix SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
  echo "config $c counters done"
done
if [ -x tools/calib/fetch_calib ]; then
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_calib_fetch -o p -- tools/calib/fetch_calib > $OUT/${TAG}_calib_fetch.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_calib_write -o p -- tools/calib/fetch_calib > $OUT/${TAG}_calib_write.log 2>&1
  echo "calibration done"
fi
echo profiles done
