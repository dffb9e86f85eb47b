#!/bin/bash
# Development aid: the headline workload through the two call patterns of bench.py (steps grouped into
# alilqr_solve_batches calls / one alilqr_solve_batch per step), several group sizes and depths.
out=gpurun_out/r3_group_probe.txt; mkdir -p gpurun_out; : > $out
run() { echo "== $*" >> $out; timeout -k 10 300 python bench.py "$@" --no-cpu 2>>gpurun_out/r3_group_probe.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print(round(d['value']), 'ms/step', round(d['ms_per_step'],2), c['entry_point'], 'steps/call', c['steps_per_call'], 'in flight', c['calls_in_flight'], 'kernel_ms', round(d['roofline']['kernel_ms'],1), 'frac', round(d['roofline']['frac'],4), 'frac_job', round(d['roofline']['frac_job'],4), 'hinted', c.get('with_order_hint') and round(c['with_order_hint']['value']))" >> $out || { tail -5 gpurun_out/r3_group_probe.err; exit 1; }; }
for a in "$@"; do run --steps 20 --warmup 5 $a; done
cat $out
