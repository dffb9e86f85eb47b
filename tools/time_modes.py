#!/usr/bin/env python3
"""Development aid: kernel time of one batch solve per solver generation / model (inputs resident, HIP events)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from ilqr_amd import batch, problems

from ilqr_amd import capi as _capi
_capi.set_tuning(lanes=int(os.environ.get("ALILQR_LANES", "0") or 0))   # development switch of this script

def main():
    B = int(os.environ.get("B", "4096"))
    sets = {
        "legacy lat H50": problems.lat_legacy(B),
        "legacy dyn H100": problems.full_legacy("dyn", B, 100, 2),
        "legacy dyn H60": problems.full_legacy("dyn", B, 60, 2),
        "legacy kin H50": problems.full_legacy("kin", B, 50, 1),
        "fast box H50": problems.fast_box(B),
        "fast box+linear H100": problems.fast_box_linear(B),
        "new_al box6 H50": problems.new_al_box(B),
        "new_al lat4 H50": problems.new_al_lat(B),
        "new_al kin+linear H50": problems.kin_linear(B),
        "new_al circle6 H50": problems.new_al_circle(B),
        "new_al kin+rect H50": problems.kin_rect(B),
        "new_al lat4 circle H30": problems.new_al_lat_circle(B),
        "new_al lat4 rect H50": problems.new_al_lat_rect(B),
    }
    only = sys.argv[1:]
    for name, prob in sets.items():
        if only and not any(o in name for o in only):
            continue
        kw = {}
        if isinstance(prob[0], tuple):
            prob, kw = prob
        ins = [torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device="cuda:0") for a in prob[1:]]
        run = lambda: batch.solve_batch(prob[0], *ins, extras=True, sync=False, **kw)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            o = run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        st = o["status"].cpu().numpy()
        print(f"{name:24s} lanes={os.environ.get('ALILQR_LANES','auto'):>4s} B={B} {ms:8.2f} ms {B / ms * 1e3:10.0f} solves/s  "
              f"inner mean {st[:,0].mean():6.1f} max {st[:,0].max():4d}  rollouts mean {st[:,3].mean():7.1f}  converged {np.mean((st[:,2] & 1) != 0):.3f}", flush=True)

if __name__ == "__main__":
    main()
