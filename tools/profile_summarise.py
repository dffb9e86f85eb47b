#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of tools/profile_round.sh into the committed summaries under profiles/.

  python tools/profile_summarise.py r03 "2 3 4a 4b 5"
writes profiles/<tag>_kernel_stats.txt (kernel trace of the default bench command), profiles/<tag>_fetch_calibration.txt
(what FETCH_SIZE / WRITE_SIZE report for known byte counts in the solve kernel's access widths) and, per configuration,
profiles/<tag>_pmc_<config>.json / .txt.  rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  The read-side correction
factor is the calibrated one for 8-byte-per-lane loads (falls back to 2, the documented factor of 16-byte loads, when no
calibration run is present).
"""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
configs = (sys.argv[2] if len(sys.argv) > 2 else "2").split()
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def find(d, pat):
    hits = glob.glob(os.path.join(G, d, "**", pat), recursive=True)
    return hits[0] if hits else None


def counters(d, match="solve_kernel"):
    """{counter: total over dispatches}, number of dispatches, of the kernels whose name holds `match`"""
    f = find(d, "*counter_collection.csv")
    agg = collections.defaultdict(float)
    launches = set()
    if f:
        for r in csv.DictReader(open(f)):
            if match in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"])
                launches.add(r.get("Dispatch_Id"))
    return agg, max(1, len(launches))


def bench_line(log):
    path = os.path.join(G, log)
    if os.path.exists(path):
        for l in open(path):
            if l.startswith("{"):
                return json.loads(l)
    return None


# ---- kernel trace of the default bench command
lines = []
f = find(tag + "_kt", "*kernel_stats.csv")
if f:
    lines.append("rocprofv3 --kernel-trace --stats -- python3 bench.py")
    lines.append("name | calls | total_ns | avg_ns | pct")
    for r in csv.DictReader(open(f)):
        lines.append(f"{r['Name']} | {r['Calls']} | {r['TotalDurationNs']} | {r['AverageNs']} | {r['Percentage']}")
    f2 = find(tag + "_kt", "*kernel_trace.csv")
    if f2:
        # the launches of the timed region are the large ones: duration per grid size
        by_grid = collections.defaultdict(list)
        first = {}
        for r in csv.DictReader(open(f2)):
            if "solve_kernel" in r["Kernel_Name"]:
                by_grid[(r["Kernel_Name"], r["Grid_Size_X"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                first.setdefault((r["Kernel_Name"], r["Grid_Size_X"]), r)
        lines.append("solve_kernel dispatches by grid size: kernel | grid | dispatches | mean duration ms")
        for (name, grid), d in sorted(by_grid.items(), key=lambda kv: -sum(kv[1])):
            r = first[(name, grid)]
            regs = " ".join(f"{k}={r[k]}" for k in ("Workgroup_Size_X", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size", "LDS_Block_Size") if k in r)
            lines.append(f"{name} | {grid} | {len(d)} | {sum(d) / len(d) / 1e6:.3f} | {regs}")
    b = bench_line(tag + "_kt.log")
    if b:
        lines.append("bench line of the profiled run: " + json.dumps(b))
    open(os.path.join(P, tag + "_kernel_stats.txt"), "w").write("\n".join(lines) + "\n")

# ---- calibration of FETCH_SIZE / WRITE_SIZE on 8-byte-per-lane accesses
calib = []
read_factor = 2.0
cf, _ = {}, 0
GiB = float(1 << 30)
per_kernel = {}
for kind, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = find(f"{tag}_calib_{kind}", "*counter_collection.csv")
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == ctr:
            k = r["Kernel_Name"].split("(")[0]
            per_kernel[(k, ctr)] = per_kernel.get((k, ctr), 0.0) + float(r["Counter_Value"]) * 1024.0
if per_kernel:
    calib.append("tools/calib/fetch_calib under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate runs); buffer 1 GiB = 1073741824 bytes")
    calib.append("kernel | counter | reported bytes | true bytes | true / reported")
    truth = {"read8": GiB, "read16": GiB, "bcast8": GiB / 32, "write8": GiB}
    for (k, ctr), v in sorted(per_kernel.items()):
        for name, t in truth.items():
            if name in k and ((ctr == "FETCH_SIZE") != (name == "write8")):
                calib.append(f"{name} | {ctr} | {v:.6g} | {t:.6g} | {t / v if v else float('nan'):.4f}")
                if name == "read8" and v:
                    read_factor = t / v
    calib.append(f"read-side factor used for the solve kernel's 8-byte-per-lane loads: {read_factor:.4f}")
    open(os.path.join(P, tag + "_fetch_calibration.txt"), "w").write("\n".join(calib) + "\n")

# ---- per configuration
sys.path.insert(0, ROOT)
import bench
for cfg in configs:
    out, summary = [], {}
    run = bench_line(f"{tag}_c{cfg}_fetch.log") or bench_line(f"{tag}_c{cfg}_stall.log")
    fetch, nf = counters(f"{tag}_c{cfg}_fetch")
    write, nw = counters(f"{tag}_c{cfg}_write")
    if not fetch and not write:
        continue
    cmd = f"python3 tools/pmc_run.py --config {cfg}"
    if fetch:
        raw = fetch["FETCH_SIZE"] / nf * 1024
        out += [f"rocprofv3 --pmc FETCH_SIZE -- {cmd}", f"FETCH_SIZE per launch {raw:.6g} bytes raw ; x {read_factor:.4f} (calibrated for 8-byte-per-lane loads) = {raw * read_factor:.6g} bytes"]
        summary["fetch_bytes_per_launch_raw"] = raw
        summary["fetch_bytes_per_launch_corrected"] = raw * read_factor
    if write:
        raw = write["WRITE_SIZE"] / nw * 1024
        out += [f"rocprofv3 --pmc WRITE_SIZE -- {cmd}", f"WRITE_SIZE per launch {raw:.6g} bytes"]
        summary["write_bytes_per_launch"] = raw
    if fetch and write:
        summary["hbm_bytes_per_launch"] = summary["fetch_bytes_per_launch_corrected"] + summary["write_bytes_per_launch"]
        # the calibration also shows that loads of ONE address by all lanes of a group (the roll-outs' reads of the
        # reference point, gains and multipliers) are reported exactly: the corrected read bytes are an upper bound
        summary["hbm_bytes_per_launch_lower_bound"] = summary["fetch_bytes_per_launch_raw"] + summary["write_bytes_per_launch"]
        out.append(f"memory-side bytes per launch: corrected reads + writes = {summary['hbm_bytes_per_launch']:.6g} (upper bound); "
                   f"raw reads + writes = {summary['hbm_bytes_per_launch_lower_bound']:.6g} (lower bound: same-address loads are reported exactly)")
    for name, what in (("stall", "wave-cycle shares"), ("mix", "executed instruction mix")):
        c, n = counters(f"{tag}_c{cfg}_{name}")
        if not c:
            continue
        out.append(f"rocprofv3 --pmc {' '.join(sorted(c))} -- {cmd}   ({what})")
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
    if run:
        summary.update(batch=run["batch"], launch_ms=run["launch_ms"], algorithmic_flops_per_launch=run["algorithmic_flops_per_launch"],
                       algorithmic_hbm_bytes_per_launch=run["algorithmic_hbm_bytes_per_launch"])
        if summary.get("hbm_bytes_per_launch"):
            summary["hbm_bytes_per_solve"] = summary["hbm_bytes_per_launch"] / run["batch"]
        if summary.get("executed_fp64_lane_flops_per_launch"):
            summary["executed_over_algorithmic_flops"] = summary["executed_fp64_lane_flops_per_launch"] / run["algorithmic_flops_per_launch"]
            out.append(f"executed / algorithmic fp64 flops = {summary['executed_over_algorithmic_flops']:.3f}")
        out.append("profiled run: " + json.dumps(run))
    summary["kernel_source_id"] = (run or {}).get("kernel_source_id", bench.kernel_source_id())
    summary["config"] = cfg
    summary["read_factor"] = read_factor
    summary["command"] = cmd + " (3 identical launches of the configuration's batch through alilqr_solve_batch; per-launch averages)"
    open(os.path.join(P, f"{tag}_pmc_{cfg}.txt"), "w").write("\n".join(out) + "\n")
    json.dump(summary, open(os.path.join(P, f"{tag}_pmc_{cfg}.json"), "w"), indent=1)
    print(f"config {cfg}:", "\n".join(out[-6:]))
print("\n".join(lines[:12] + calib))
