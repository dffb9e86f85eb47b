#!/usr/bin/env python3
"""Development aid: build a variant of the HIP library with extra compiler flags next to the product one.

  python tools/build_variant.py <name> [-DFLAG ...] [--jobs N]   ->  ilqr_amd/csrc/libalilqr_hip_<name>.so
e.g.  python tools/build_variant.py debug -DALILQR_DEBUG_BOUNDS      (bounds-checked kernels, tools/bounds_check.py)
Load it with ALILQR_LIB=<path> (ilqr_amd.capi).
"""
import concurrent.futures, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge


def main():
    name = sys.argv[1]
    jobs = 4
    flags = []
    it = iter(sys.argv[2:])
    for a in it:
        if a == "--jobs":
            jobs = int(next(it))
        else:
            flags.append(a)
    obj = os.path.join(ge.CSRC, "obj_" + name)
    os.makedirs(obj, exist_ok=True)
    units = [(os.path.join(obj, "api.o"), os.path.join(ge.CSRC, "alilqr_api.hip"), [])]
    for model, mode in ge._families():
        units.append((os.path.join(obj, f"family_m{model}_k{mode}.o"), os.path.join(ge.CSRC, "solve_family.hip"),
                      [f"-DALILQR_FAMILY_MODEL={model}", f"-DALILQR_FAMILY_MODE={mode}"]))
    with concurrent.futures.ThreadPoolExecutor(jobs) as pool:
        list(pool.map(lambda u: subprocess.run(["/opt/rocm/bin/hipcc"] + ge.HIP_FLAGS + flags + u[2] + ["-c", u[1], "-o", u[0]],
                                               check=True, cwd=ge.CSRC), units))
    out = os.path.join(ge.CSRC, f"libalilqr_hip_{name}.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + [u[0] for u in units], check=True)
    print(out)


if __name__ == "__main__":
    main()
