#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters of the solve kernel from a counter_collection.csv."""
import collections, csv, sys
agg = collections.defaultdict(float); n = 0
for r in csv.DictReader(open(sys.argv[1])):
    if 'solve_kernel' in r['Kernel_Name']:
        agg[r['Counter_Name']] += float(r['Counter_Value'])
for k, v in sorted(agg.items()):
    print(f"{k} {v:.6g}")
if 'SQ_WAVE_CYCLES' in agg:
    w = agg['SQ_WAVE_CYCLES']
    for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_ANY'):
        if k in agg: print(f"{k}/SQ_WAVE_CYCLES = {agg[k]/w:.3f}")
