#!/usr/bin/env python3
"""Development aid: GPU solve vs CPU oracle on the golden problem sets (prints, no asserts)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu, oracle_lib as ol
from ilqr_amd import batch

from ilqr_amd import capi as _capi
_capi.set_tuning(lanes=int(os.environ.get("ALILQR_LANES", "0") or 0))   # development switch of this script

def main():
    sets = sys.argv[1:] or ["lat_legacy", "dyn_legacy", "kin_legacy", "fast_box", "fast_linear"]
    for name in sets:
        g = gu.load(name)
        desc, x0, goals, Q, R, cons = gu.problem_from_golden(g)
        ref = ol.solve_batch(desc, x0, goals, Q, R, cons)
        t = time.time()
        out = batch.solve_batch(desc, x0, goals, Q, R, cons)
        dt = time.time() - t
        o = {k: v.cpu().numpy() for k, v in out.items()}
        bad = 0
        for i in range(desc.batch):
            ex = gu.rel_err(o["x"][i], ref["x"][i]); eu = gu.rel_err(o["u"][i], ref["u"][i])
            en = gu.rel_err(o["node_x"][i], ref["node_x"][i])
            ec = abs(o["cost"][i] - ref["cost"][i]) / max(1, abs(ref["cost"][i]))
            same = tuple(o["status"][i]) == tuple(ref["status"][i])
            worst = max(ex, eu, en, ec)
            if worst > 1e-6 or not np.isfinite(worst):
                bad += 1
                print(f"  {name}[{i}] stable={bool(g['stable'][i])} gpu status {o['status'][i]} oracle {ref['status'][i]} "
                      f"ex {ex:.2e} eu {eu:.2e} en {en:.2e} ec {ec:.2e} cost {o['cost'][i]:.8g} / {ref['cost'][i]:.8g}")
        xi = gu.rel_err(o["x_init"], ref["x_init"])
        print(f"{name}: B={desc.batch} lanes={os.environ.get('ALILQR_LANES','auto')} gpu {dt*1e3:.1f} ms  mismatches(>1e-6) {bad}  x_init err {xi:.2e}", flush=True)

if __name__ == "__main__":
    main()
