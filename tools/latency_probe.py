#!/usr/bin/env python3
"""Development aid: latency of the longest problems run alone (one wave) vs the whole batch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ilqr_amd import batch, problems

def timed(desc, x0, goals, Q, R, cons, reps=3):
    s = batch.BatchSolver(desc, x0, goals, Q, R, cons)
    s.run(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); o = s.run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts), o["status"].cpu().numpy()

B = 4096
desc, x0, goals, Q, R, cons = problems.lat_legacy(B, 50, seed=0)
t_all, st = timed(desc, x0, goals, Q, R, cons)
order = np.argsort(-st[:, 0].astype(np.int64) * 100000 - st[:, 3])
print(f"batch {B}: {t_all:.1f} ms; inner iters mean {st[:,0].mean():.1f} max {st[:,0].max()}; rollouts mean {st[:,3].mean():.0f} max {st[:,3].max()}")
h = np.bincount(np.minimum(st[:, 0] // 50, 8)); print("inner-iteration histogram (bins of 50):", h)
for n in (2, 64, 512):
    idx = order[:n]
    d2, x2, *_ = problems.lat_legacy(n, 50, x0=x0[idx])
    t, s2 = timed(d2, x2, goals, Q, R, cons)
    print(f"longest {n:4d} problems alone: {t:.1f} ms  ({t*1e3/s2[:,0].max():.1f} us per inner iteration of the longest, {s2[:,0].max()} iterations, {s2[:,3].max()} rollouts)")
idx = order[-512:]
d2, x2, *_ = problems.lat_legacy(512, 50, x0=x0[idx])
t, s2 = timed(d2, x2, goals, Q, R, cons)
print(f"shortest 512 problems alone: {t:.1f} ms (max {s2[:,0].max()} iterations)")
