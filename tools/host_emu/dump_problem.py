#!/usr/bin/env python3
"""Write one synthetic problem set as the flat file tools/host_emu/driver.cpp reads: descriptor, x0, goals, Q, R, cons."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ilqr_amd import problems

def main():
    name, B, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    sets = {"new_box": lambda: problems.new_al_box(B), "new_lat": lambda: problems.new_al_lat(B),
            "new_box_active": lambda: problems.new_al_box(B, y_max=20.0),
            "lat_legacy": lambda: problems.lat_legacy(B), "fast_box": lambda: problems.fast_box(B),
            "dyn_legacy": lambda: problems.full_legacy("dyn", B, 60, 3),
            "circle6": lambda: problems.new_al_circle(B), "lat_circle": lambda: problems.new_al_lat_circle(B)}
    kw = {}
    if name.startswith("golden:"):
        # the first B problems of a golden set, e.g. golden:lon_legacy
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import golden_util as gu
        desc, x0, goals, Q, R, cons = gu.problem_from_golden(gu.load(name[7:]), idx=list(range(B)))
    elif name in ("lat_rect", "kin_rect"):
        (desc, x0, goals, Q, R, cons), kw = (problems.new_al_lat_rect if name == "lat_rect" else problems.kin_rect)(B)
    else:
        desc, x0, goals, Q, R, cons = sets[name]()
    with open(out, "wb") as f:
        f.write(bytes(desc))
        for a in (x0, goals, Q, R, cons):
            f.write(np.ascontiguousarray(a, dtype=np.float64).tobytes())
        if kw:
            from ilqr_amd import capi
            lo, ro = capi.pack_rectangles(kw["left_obs"]), capi.pack_rectangles(kw["right_obs"])
            f.write(np.array([len(lo), len(ro), kw["first_row"], 0], dtype=np.int32).tobytes())
            f.write(lo.tobytes()); f.write(ro.tobytes())

if __name__ == "__main__":
    main()
