#!/bin/bash
# Profiles of the bench command on the GPU box: kernel trace + three counter passes (each its own run).
# Usage (on the box, from the repo root):  bash tools/profile_round.sh r01
# Raw output goes to gpurun_out/<tag>_*, summaries are written by tools/profile_summarise.py afterwards.
set -e
TAG=${1:-r01}
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_kt -o kt -- python3 bench.py > $OUT/${TAG}_kt.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu > $OUT/${TAG}_pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu > $OUT/${TAG}_pmc_write.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/${TAG}_pmc_sq -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu > $OUT/${TAG}_pmc_sq.log 2>&1
bash tools/pmc_extra.sh $TAG > $OUT/${TAG}_x_summary.txt 2>&1
echo profiles done
