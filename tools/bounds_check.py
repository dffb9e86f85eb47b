#!/usr/bin/env python3
"""Development aid: run every kernel family, at several lane-group widths, through the bounds-checked build
(python tools/build_variant.py debug -DALILQR_DEBUG_BOUNDS) and print the violation bits of each solve.
On the GPU box:  ALILQR_LIB=$PWD/ilqr_amd/csrc/libalilqr_hip_debug.so python tools/bounds_check.py
"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ilqr_amd import batch, capi, problems


def main():
    lib = capi.load_library()
    lib.alilqr_debug_read.restype = C.c_int32
    cases = {
        "legacy lat": (problems.lat_legacy(160, 50, seed=3), {}),
        "legacy lat H=200": (problems.lat_legacy(6, 200, seed=21), {}),
        "legacy dyn H=100": (problems.full_legacy("dyn", 64, 100, 2), {}),
        "legacy kin": (problems.full_legacy("kin", 48, 50, 5), {}),
        "fast box": (problems.fast_box(48), {}),
        "fast box+linear H=100": (problems.fast_box_linear(32), {}),
        "new_al box6": (problems.new_al_box(64), {}),
        "new_al box6 active bound": (problems.new_al_box(48, y_max=20.0), {}),
        "new_al lat4": (problems.new_al_lat(48), {}),
        "new_al circle6": (problems.new_al_circle(48), {}),
        "new_al circle4": (problems.new_al_lat_circle(48), {}),
        "new_al lat rect": problems.new_al_lat_rect(32),
        "new_al kin rect (config 3)": problems.kin_rect(64),
        # large batches with diverging solves (the far-argument trig calls run): the 6-state FAST / NEW_AL kernels at
        # the batch and widths a timing run uses
        "fast box B=4096": (problems.fast_box(4096), {}),
        "fast box+linear stiff B=2048 H=100": (problems.fast_box_linear(2048, stiff=True), {}),
        "new_al box6 B=4096": (problems.new_al_box(4096), {}),
        "new_al kin rect B=4096": problems.kin_rect(4096),
        "legacy kin H=50 (diverging) B=2048": (problems.full_legacy("kin", 2048, 50, 1), {}),
    }
    big = {k for k in cases if " B=" in k}
    # per-node tables and per-problem goal lists (the table / goal index checks), fewer resident groups than problems
    d0, x0, goals, Q, R, cons = problems.lat_legacy(40, 50, seed=4)
    d2 = capi.make_problem("lat", "legacy", 50, 40, 1.0, 0.1, v=10.0, goal_batch=40, table_nodes=51)
    cases["legacy lat per-node tables, per-problem goals"] = ((d2, x0, np.tile(goals, (40, 1, 1)), np.tile(Q, (51, 1, 1)),
                                                               np.tile(R, (51, 1, 1)), np.tile(cons, (51, 1))), {})
    bad = 0
    for name, (prob, kw) in cases.items():
        for lanes in ((0, 8, 16) if name in big else (0, 1, 4, 8, 16, 32)):
            capi.set_tuning(lanes=lanes)
            out = batch.solve_batch(*prob, **kw)
            word = C.c_uint(0)
            assert lib.alilqr_debug_read(C.byref(word)) == 0
            fin = float(np.isfinite(out["cost"].cpu().numpy()).mean())
            print(f"{name:28s} lanes={lanes or 'auto':>4} violations=0x{word.value:04x} finite={fin:.2f}", flush=True)
            bad += word.value != 0
    capi.set_tuning()
    print("TOTAL solves with violations:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
