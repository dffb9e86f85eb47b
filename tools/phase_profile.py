#!/usr/bin/env python3
"""Development aid: per-phase cycle breakdown of the solve kernel.

Builds a profiling variant of the library (-DALILQR_PROF: s_memtime stamps at
phase boundaries, summed over waves) next to the product one and runs the
bench workload once.  Run on the GPU box:  python tools/phase_profile.py [lanes]
"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

def build():
    out = os.path.join(ROOT, "ilqr_amd", "csrc", "libalilqr_hip_prof.so")
    src = os.path.join(ROOT, "ilqr_amd", "csrc", "alilqr_api.hip")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(os.path.join(ROOT, "ilqr_amd", "csrc", f)) for f in ("alilqr_api.hip", "solver.hpp", "models.hpp", "launch.hpp", "families.inc")):
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                        "-DALILQR_PROF", "-DALILQR_SINGLE_TU", "-DALILQR_HEADLINE_ONLY=0", "-o", out, src], check=True)
    return out

def main():
    # PROF_LIB: another profiling build (e.g. one linked from per-family objects compiled with -DALILQR_PROF)
    path = os.environ.get("PROF_LIB") or build()
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        return
    from ilqr_amd import capi, batch, problems
    capi.LIB_PATH = path
    lib = capi.load_library()
    capi.set_tuning(lanes=int(os.environ.get("ALILQR_LANES", "0") or 0))   # development switch of this script
    import torch
    B = int(os.environ.get("B", "4096"))
    workloads = {"lat_legacy": lambda: problems.lat_legacy(B, 50, seed=0),
                 "dyn_legacy_h100": lambda: problems.full_legacy("dyn", B, 100, 2),
                 "fast_box": lambda: problems.fast_box(B), "new_al_box6": lambda: problems.new_al_box(B),
                 "fast_box_linear_h100": lambda: problems.fast_box_linear(B), "kin_rect": lambda: problems.kin_rect(B),
                 "new_al_circle6": lambda: problems.new_al_circle(B), "kin_linear": lambda: problems.kin_linear(B),
                 "new_al_lat4": lambda: problems.new_al_lat(B)}
    wl = os.environ.get("WORKLOAD", "lat_legacy")
    prob, kw = workloads[wl](), {}
    if isinstance(prob[0], tuple):
        prob, kw = prob
    desc = prob[0]
    s = batch.BatchSolver(*prob, **kw)
    s.run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); o = s.run(); e1.record(); torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    assert lib.alilqr_prof_read(buf) == 0
    v = np.array(list(buf), dtype=np.float64)
    names = ["pull+init", "outer-start cost", "P1 linearise", "P2 riccati", "P3 rollouts", "P4 bookkeeping", "-", "group 0 idle, others busy"]
    waves = v[8]
    tot = v[:8].sum()
    st = o["status"].cpu().numpy()
    print(f"{wl} lanes={os.environ.get('ALILQR_LANES','auto')} B={B} kernel {e0.elapsed_time(e1):.1f} ms  waves {int(waves)}  mean wave busy {tot/waves/1e6:.1f} Mcycles  mean inner {st[:,0].mean():.1f}")
    for n, x in zip(names, v[:8]):
        print(f"  {n:26s} {100*x/tot:6.2f} %   {x/waves/1e6:8.2f} Mcyc/wave")
    if v[10] > 0:
        print(f"  roll-out passes {int(v[10])}: {v[9]/v[10]:.0f} cycles per pass inside the roll-out, {(v[4]-v[9])/v[10]:.0f} in the rest of P3")
    if v[11:16].sum() > 0:
        names3 = ["P3 start -> first pass", "pass: before the roll-out", "pass: roll-out", "pass: after the roll-out", "last pass -> P3 end"]
        print("  P3 in parts (Mcyc/wave): " + "; ".join(f"{n} {x/waves/1e6:.2f}" for n, x in zip(names3, v[11:16])))

if __name__ == "__main__":
    main()
