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
    }
    bad = 0
    for name, (prob, kw) in cases.items():
        for lanes in (0, 1, 4, 8, 16, 32):
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
