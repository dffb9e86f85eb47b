#!/bin/bash
# Development aid, on the GPU box: GPU tests, then the round's profiles, their summary, then one bench line per configuration
# (the lines read the summarised counters).  Afterwards, locally:  python tools/profile_summarise.py <tag> "<configs>"  and
# copy gpurun_out/bench_all.txt to profiles/<tag>_bench_all_configs.txt.
set -e
TAG=${1:-r03}; CONFIGS=${2:-"2 3 4a 4b 5"}
mkdir -p gpurun_out
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { tail -5 gpurun_out/gpu_tests.log; exit 1; }
tail -1 gpurun_out/gpu_tests.log
bash tools/profile_round.sh $TAG "$CONFIGS"
python tools/profile_summarise.py $TAG "$CONFIGS" > gpurun_out/summarise.log 2>&1
: > gpurun_out/bench_all.txt
for c in $CONFIGS; do
  timeout -k 10 400 python bench.py --config $c 2>/dev/null | grep '^{' >> gpurun_out/bench_all.txt
  echo "bench $c done"
done
