#!/usr/bin/env python3
"""Profiling driver: N identical launches of one benchmark configuration through alilqr_solve_batch, nothing else on
the GPU, so that every solve_kernel dispatch of a rocprofv3 pass is the same launch.

  python3 tools/pmc_run.py --config 2 [--batch B] [--launches 3]
Prints one JSON line: configuration, batch, launches, mean launch time (HIP events), iteration / roll-out totals of a
launch and its algorithmic operation / byte counts (the figures bench.py's roofline uses).
"""
import argparse, ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from ilqr_amd import batch, capi


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="2", choices=sorted(bench.CONFIGS))
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--launches", type=int, default=3)
    a = ap.parse_args()
    B = a.batch or bench.CONFIGS[a.config]["batch"]
    prob, kw = bench.make_workload(a.config, B, 0)
    # the lane width of the bench's library call: configurations without optional inputs go through alilqr_solve_batches,
    # 20 batches per call, and the library picks the width for the call's total
    if not kw and a.config not in ("3", "5"):
        call = type(prob[0]).from_buffer_copy(prob[0]); call.batch = B * 20
        lanes = capi.load_library().alilqr_lanes_for(C.byref(call))
        assert lanes > 0, capi.last_error()
        capi.set_tuning(lanes=lanes)
    sv = batch.BatchSolver(*prob, extras=False, **kw)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.launches)]
    for e0, e1 in ev:
        e0.record(); sv.run(); e1.record()
        torch.cuda.synchronize()
    st = sv.out["status"].cpu().numpy()
    d = prob[0]
    print(json.dumps(dict(config=a.config, batch=B, launches=a.launches, launch_ms=[e0.elapsed_time(e1) for e0, e1 in ev],
                          inner_iterations=int(st[:, 0].sum()), rollouts=int(st[:, 3].sum()),
                          algorithmic_flops_per_launch=bench.algorithmic_flops(st, d.horizon, d.n),
                          algorithmic_hbm_bytes_per_launch=bench.algorithmic_bytes(d),
                          kernel_source_id=bench.kernel_source_id())), flush=True)


if __name__ == "__main__":
    main()
