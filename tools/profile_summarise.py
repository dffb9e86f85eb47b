#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of tools/profile_round.sh into the committed summaries under profiles/.

  python tools/profile_summarise.py r01
writes profiles/<tag>_kernel_stats.txt, profiles/<tag>_pmc.txt and profiles/<tag>_pmc.json
(hbm_bytes_per_launch = 2 x FETCH_SIZE + WRITE_SIZE, both reported in KiB by rocprofv3; the factor 2 is the
gfx950 correction for FETCH_SIZE counting 128-byte requests as 64 bytes).
"""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def find(d, pat):
    hits = glob.glob(os.path.join(G, d, "**", pat), recursive=True)
    return hits[0] if hits else None


def counters(d):
    f = find(d, "*counter_collection.csv")
    agg = collections.defaultdict(float)
    launches = set()
    if f:
        for r in csv.DictReader(open(f)):
            if "solve_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"])
                launches.add(r.get("Dispatch_Id"))
    return agg, max(1, len(launches))


lines = []
f = find(tag + "_kt", "*kernel_stats.csv")
if f:
    lines.append("rocprofv3 --kernel-trace --stats -- python3 bench.py")
    lines.append("name | calls | total_ns | avg_ns | pct")
    for r in csv.DictReader(open(f)):
        lines.append(f"{r['Name']} | {r['Calls']} | {r['TotalDurationNs']} | {r['AverageNs']} | {r['Percentage']}")
    f2 = find(tag + "_kt", "*kernel_trace.csv")
    if f2:
        for r in csv.DictReader(open(f2)):
            if "solve_kernel" in r["Kernel_Name"]:
                lines.append("dispatch: " + " ".join(f"{k}={r[k]}" for k in ("Kernel_Name", "Grid_Size_X", "Workgroup_Size_X", "VGPR_Count",
                                                                             "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size", "LDS_Block_Size") if k in r))
                break
    log = os.path.join(G, tag + "_kt.log")
    if os.path.exists(log):
        for l in open(log):
            if l.startswith("{"):
                lines.append("bench line of the profiled run: " + l.strip())
    open(os.path.join(P, tag + "_kernel_stats.txt"), "w").write("\n".join(lines) + "\n")

out = []
fetch, nf = counters(tag + "_pmc_fetch")
write, nw = counters(tag + "_pmc_write")
sq, ns = counters(tag + "_pmc_sq")
summary = {}
if fetch:
    kb = fetch["FETCH_SIZE"] / nf
    out += ["rocprofv3 --pmc FETCH_SIZE -- python3 bench.py --steps 1 --warmup 0 --no-cpu",
            f"FETCH_SIZE per launch {kb:.6g} KiB raw ; x2 (gfx950 correction) = {2 * kb * 1024:.6g} bytes"]
    summary["fetch_bytes_per_launch_corrected"] = 2 * kb * 1024
if write:
    kb = write["WRITE_SIZE"] / nw
    out += ["rocprofv3 --pmc WRITE_SIZE -- python3 bench.py --steps 1 --warmup 0 --no-cpu",
            f"WRITE_SIZE per launch {kb:.6g} KiB = {kb * 1024:.6g} bytes"]
    summary["write_bytes_per_launch"] = kb * 1024
if fetch and write:
    summary["hbm_bytes_per_launch"] = summary["fetch_bytes_per_launch_corrected"] + summary["write_bytes_per_launch"]
    out.append(f"hbm_bytes_per_launch = {summary['hbm_bytes_per_launch']:.6g}")
if sq:
    out.append("rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD -- python3 bench.py --steps 1 --warmup 0 --no-cpu")
    for k, v in sorted(sq.items()):
        out.append(f"{k} {v / ns:.6g}")
        summary[k] = v / ns
    w = sq.get("SQ_WAVE_CYCLES")
    if w:
        out.append(f"wait share SQ_WAIT_ANY/SQ_WAVE_CYCLES = {sq['SQ_WAIT_ANY'] / w:.3f} ; VALU-active share = {sq['SQ_ACTIVE_INST_VALU'] / w:.3f}")
# extra passes of tools/pmc_extra.sh: issue stalls, instruction cache, executed instruction mix
for name, what in (("stall", "wave-cycle breakdown"), ("icache", "instruction cache"), ("mix", "executed VALU instruction mix")):
    c, n = counters(tag + "_x_" + name)
    if not c:
        continue
    out.append(f"rocprofv3 --pmc {' '.join(sorted(c))} -- python3 bench.py --steps 1 --warmup 0 --no-cpu   ({what})")
    for k, v in sorted(c.items()):
        out.append(f"{k} {v / n:.6g}")
        summary[k] = v / n
    if name == "stall" and c.get("SQ_WAVE_CYCLES"):
        w = c["SQ_WAVE_CYCLES"]
        out.append("shares of wave cycles: " + " ; ".join(f"{k} {c[k] / w:.3f}" for k in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY") if k in c))
    if name == "mix":
        fl = (2 * c.get("SQ_INSTS_VALU_FMA_F64", 0) + c.get("SQ_INSTS_VALU_ADD_F64", 0) + c.get("SQ_INSTS_VALU_MUL_F64", 0)) * 64 / n
        summary["executed_fp64_lane_flops_per_launch"] = fl
        out.append(f"executed fp64 flops per launch, all 64 lanes of every issued instruction counted = {fl:.6g}")
if out:
    # which build and workload the counters belong to: bench.py reports them only for that build
    sys.path.insert(0, ROOT)
    import bench
    summary["kernel_source_id"] = bench.kernel_source_id()
    summary["config"] = "2"
    summary["command"] = "python3 bench.py --steps 1 --warmup 0 --no-cpu (14 launches of the headline batch: 12 priming, 1 timed, 1 alone; per-launch averages)"
    open(os.path.join(P, tag + "_pmc.txt"), "w").write("\n".join(out) + "\n")
    json.dump(summary, open(os.path.join(P, tag + "_pmc.json"), "w"), indent=1)
print("\n".join(lines + out))
