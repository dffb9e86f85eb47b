"""Host logic of bench.py (no GPU): the algorithmic operation / byte counts behind the roofline fields, the workload
table, and that the committed counter summary belongs to the kernel sources in the tree."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_algorithmic_flops_follow_the_status_words():
    # two lateral problems, H = 50: 3 + 5 inner iterations, 10 + 30 roll-outs
    status = np.array([[3, 1, 1, 10], [5, 2, 2, 30]], dtype=np.int32)
    f = bench.FLOPS[4]
    want = 8 * 50 * (f["lin"] + f["quad"] + f["bwd"]) + 40 * (50 * f["dyn"] + 51 * f["cost"])
    assert bench.algorithmic_flops(status, 50, 4) == want
    # the figures of the 4-state model: backward 469, linearise 60 + 4 transcendentals, quadratise 40,
    # dynamics 30 + 6 transcendentals, cost 60 per node
    assert (f["bwd"], f["lin"], f["quad"], f["dyn"], f["cost"]) == (469, 64, 40, 36, 60)


def test_algorithmic_bytes_are_inputs_plus_outputs():
    desc = bench.make_workload("2", 8, 0)[0][0]
    # x0 + x + u + cost (doubles) + four status words per problem
    assert bench.algorithmic_bytes(desc) == 8.0 * 8 * (4 + 51 * 4 + 50 * 1 + 1) + 16.0 * 8
    # the headline batch: 2.1 KB per solve
    assert abs(bench.algorithmic_bytes(bench.make_workload("2", 4096, 0)[0][0]) - 8552448.0) < 1.0


def test_workloads_have_the_configured_shapes():
    for cfg, (n, m, H) in {"2": (4, 1, 50), "3": (6, 2, 50), "4a": (6, 2, 100), "4b": (6, 2, 100)}.items():
        prob, kw = bench.make_workload(cfg, 4, 0)
        d = prob[0]
        assert (d.n, d.m, d.horizon, d.batch) == (n, m, H, 4), cfg
    prob, kw = bench.make_workload("5", 2 * bench.STARTS, 0)
    assert prob[0].batch == 2 * bench.STARTS and kw["u_offset"].shape[0] == 2 * bench.STARTS


def test_committed_counters_are_reported_only_for_the_kernel_sources_in_the_tree():
    """bench.py attaches the committed rocprofv3 counter summary of a configuration to its line only when the summary
    was measured on the kernel sources in the tree (kernel_source_id); a summary of other sources is not reported."""
    import glob
    import json
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_pmc*.json")))
    assert files, "no counter summary is committed under profiles/"
    for path in files:
        with open(path) as fh:
            c = json.load(fh)
        cfg = str(c.get("config"))
        got = bench.measured_counters(cfg)
        if c.get("kernel_source_id") == bench.kernel_source_id():
            assert got is not None and got["kernel_source_id"] == bench.kernel_source_id()
            assert got["hbm_bytes_per_launch"] == got["fetch_bytes_per_launch_corrected"] + got["write_bytes_per_launch"]
            assert got["hbm_bytes_per_launch"] > 0
        else:
            assert got is None or got["source"] != "profiles/" + os.path.basename(path)
    assert bench.measured_counters("no such configuration") is None


def test_cpu_baseline_runs_every_configuration_on_a_small_sample():
    for cfg in ("2", "3", "4a", "4b"):
        b = bench.cpu_baseline(cfg, 4)
        assert b["kind"] == "port" and b["value"] > 0 and b["cores"] >= 1 and f"configuration {cfg}" in b["sample"]


def test_bench_refuses_to_run_without_a_device():
    import torch
    if torch.cuda.is_available():
        return
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "needs a HIP device" in (r.stderr + r.stdout)
