#!/bin/bash
# Development aid: throughput of library variants side by side (bench.py lines reduced to one number each).
#   bash tools/ab_bench.sh "<variant names, '' = the product library>" "<configs>" [repeats]
# variants are ilqr_amd/csrc/libalilqr_hip_<name>.so built by tools/build_variant.py
variants=${1:-""}; configs=${2:-"2 3"}; reps=${3:-1}
out=gpurun_out/ab_$(date +%H%M%S).txt; mkdir -p gpurun_out; : > $out
for rep in $(seq $reps); do for c in $configs; do for v in product $variants; do
  lib=""; [ $v != product ] && lib="ilqr_amd/csrc/libalilqr_hip_$v.so"
  steps=6; warm=2; [ $c = 2 ] && steps=20 && warm=5
  ALILQR_LIB=$lib timeout -k 10 300 python bench.py --config $c --steps $steps --warmup $warm --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg$c $v', round(d['value']), 'lone_ms', round(d['config']['lone_launch_ms'],1), 'light_us', round(d['config']['light_load']['per_iter_latency_us'],1))" >> $out || exit 1
done; done; done
cat $out
