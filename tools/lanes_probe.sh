#!/bin/bash
# Development aid: throughput of a configuration at fixed lane-group widths (bench.py lines reduced to one number each).
out=gpurun_out/r3_lanes_probe.txt; mkdir -p gpurun_out; : > $out
run() { echo "== $*" >> $out; timeout -k 10 400 python bench.py "$@" --no-cpu 2>>gpurun_out/r3_lanes_probe.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print(round(d['value']), 'hinted', c.get('with_order_hint') and round(c['with_order_hint']['value']), 'ms/step', round(d['ms_per_step'],2), c['entry_point'], 'lone_ms', round(c['lone_launch_ms'],1))" >> $out || { tail -5 gpurun_out/r3_lanes_probe.err; exit 1; }; }
for a in "$@"; do run $a; done
cat $out
