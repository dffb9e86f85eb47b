#!/usr/bin/env python3
"""Development aid: where the GPU's line-search decisions first leave the reference's (golden trace) and the oracle's."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from ilqr_amd import batch
import golden_util as gu, oracle_lib as ol

name = sys.argv[1] if len(sys.argv) > 1 else "fast_box"
g = gu.load(name)
prob = gu.problem_from_golden(g)
cap = 512
o = batch.solve_batch(*prob, trace=cap)
gc, gt, st = o["iter_cost"].cpu().numpy(), o["trials"].cpu().numpy(), o["status"].cpu().numpy()
oc, ot = ol.solve_batch_traced(*prob, capacity=cap)
rc, rt = g["iter_cost"], g["trials"]
np.set_printoptions(precision=17, linewidth=200)
for i in range(prob[0].batch):
    nr = int(np.isfinite(rt[i]).sum()); ng = int(st[i, 0]); no = len(oc[i])
    L = min(nr, ng, no)
    dg = np.nonzero(gt[i, :L] != rt[i, :L])[0]; do = np.nonzero(ot[i][:L] != rt[i, :L])[0]
    fg = int(dg[0]) if len(dg) else L; fo = int(do[0]) if len(do) else L
    print(f"#{i} iters ref {nr} oracle {no} gpu {ng} | first decision difference vs ref: gpu {fg} oracle {fo}")
    k = min(fg, L - 1)
    for j in range(max(0, k - 2), min(L, k + 2)):
        print(f"    it {j}: ref {rc[i, j]!r} t{int(rt[i, j])} | oracle {oc[i][j]!r} t{ot[i][j]} | gpu {gc[i, j]!r} t{gt[i, j]}")
