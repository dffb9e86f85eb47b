#!/usr/bin/env python3
"""Development aid: NEW_AL generation, GPU solve vs CPU oracle on synthetic problem sets (prints, no asserts)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu, oracle_lib as ol
from ilqr_amd import problems

from ilqr_amd import capi as _capi
_capi.set_tuning(lanes=int(os.environ.get("ALILQR_LANES", "0") or 0))   # development switch of this script

def sets(B):
    return {"new_box": problems.new_al_box(B), "new_box_active": problems.new_al_box(B, y_max=20.0),
            "new_lat": problems.new_al_lat(B), "new_box_h30": problems.new_al_box(B, H=30, seed=9)}

def main():
    B = int(os.environ.get("B", "64"))
    cpu_only = "--cpu" in sys.argv
    if not cpu_only:
        from ilqr_amd import batch
    only = [a for a in sys.argv[1:] if not a.startswith("--")]
    for name, (desc, x0, goals, Q, R, cons) in sets(B).items():
        if only and name not in only:
            continue
        t = time.time()
        ref = ol.solve_batch(desc, x0, goals, Q, R, cons)
        tc = time.time() - t
        st = ref["status"]
        print(f"{name}: oracle {tc*1e3:.0f} ms inner mean {st[:,0].mean():.1f} max {st[:,0].max()} outer mean {st[:,1].mean():.1f} "
              f"trials mean {st[:,3].mean():.1f} flags {np.bincount(st[:,2], minlength=8)} viol max {ref['viol'].max():.2e} cost[0] {ref['cost'][0]:.6f}", flush=True)
        if cpu_only:
            continue
        t = time.time()
        out = batch.solve_batch(desc, x0, goals, Q, R, cons)
        dt = time.time() - t
        o = {k: v.cpu().numpy() for k, v in out.items()}
        bad = 0
        for i in range(desc.batch):
            ex = gu.rel_err(o["x"][i], ref["x"][i]); eu = gu.rel_err(o["u"][i], ref["u"][i])
            ec = abs(o["cost"][i] - ref["cost"][i]) / max(1, abs(ref["cost"][i]))
            worst = max(ex, eu, ec)
            if worst > 1e-6 or not np.isfinite(worst):
                bad += 1
                if bad <= 8:
                    print(f"  {name}[{i}] gpu status {o['status'][i]} oracle {ref['status'][i]} ex {ex:.2e} eu {eu:.2e} ec {ec:.2e} "
                          f"cost {o['cost'][i]:.8g} / {ref['cost'][i]:.8g}")
        same = (o["status"] == ref["status"]).all(axis=1).mean()
        xi = gu.rel_err(o["x_init"], ref["x_init"])
        print(f"{name}: B={desc.batch} lanes={os.environ.get('ALILQR_LANES','auto')} gpu {dt*1e3:.1f} ms  mismatches(>1e-6) {bad}  "
              f"same counts {same:.3f}  x_init err {xi:.2e}", flush=True)

if __name__ == "__main__":
    main()
