#!/usr/bin/env python3
"""Development aid: per-iteration cost / line-search traces of the GPU solve against the CPU oracle's, per problem set
(how long the two agree before the first differing line-search decision).  Runs on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from ilqr_amd import batch, problems
import oracle_lib as ol


def main():
    sets = {
        "lat_legacy": problems.lat_legacy(256, 50, seed=0),
        "dyn_legacy_h100": problems.full_legacy("dyn", 32, 100, 2),
        "kin_legacy": problems.full_legacy("kin", 32, 50, 1),
        "fast_box": problems.fast_box(32),
        "fast_box_linear_h100": problems.fast_box_linear(32),
        "new_al_box6_active": problems.new_al_box(48, y_max=20.0),
        "new_al_circle4": problems.new_al_lat_circle(48),
        "new_al_circle6": problems.new_al_circle(48),
    }
    for name, prob in sets.items():
        cap = 512
        o = batch.solve_batch(*prob, trace=cap)
        gc, gt, st = o["iter_cost"].cpu().numpy(), o["trials"].cpu().numpy(), o["status"].cpu().numpy()
        oc, ot = ol.solve_batch_traced(*prob, capacity=cap)
        fr, worst, full = [], [], 0
        for i in range(prob[0].batch):
            ng, no = min(cap, int(st[i, 0])), len(oc[i])
            L = min(ng, no)
            diff = np.nonzero(gt[i, :L] != ot[i][:L])[0]
            if len(diff):
                L = int(diff[0])
            elif ng == no:
                full += 1
            with np.errstate(invalid="ignore", over="ignore"):
                e = np.abs(gc[i, :L] - oc[i][:L]) / np.maximum(1.0, np.abs(oc[i][:L]))
            e = np.where(np.isfinite(e), e, np.where(np.isfinite(oc[i][:L]) == np.isfinite(gc[i, :L]), 0.0, np.inf))
            worst.append(e.max() if L else 0.0)
            fr.append(L / max(1, no))
        worst, fr = np.array(worst), np.array(fr)
        print(f"{name:24s} B={prob[0].batch:4d} identical decision sequence {full:4d} | prefix share min {fr.min():.2f} median {np.median(fr):.2f} "
              f"| worst prefix err: max {worst.max():.1e} 99% {np.quantile(worst, 0.99):.1e} median {np.median(worst):.1e} | >1e-9: {(worst > 1e-9).sum()}", flush=True)


if __name__ == "__main__":
    main()
