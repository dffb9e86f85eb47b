#!/usr/bin/env python3
"""Throughput of the batched AL-iLQR solve path on MI355X.

One "step" = one complete solve of the workload batch.  The default workload is the configuration BASELINE.json
quotes the solves/sec metric on: 4096 lateral-bicycle problems (H=50, box-constrained, start states randomised
with seed = rank) per GPU, weak scaling.  Inputs are resident in HBM before the timed region; problems never
exchange data, so the ranks shard the batch with no collective inside the solve; one gather of the final cost
vectors over RCCL at the end of the K steps is the only exchange.

Consecutive steps are independent batches (each with start states and output buffers of its own).  A batch ends
when its longest solve ends, so batches solved one call at a time leave the SIMDs idle during every tail.  Two ways
to keep the machine full:
  --group G (default 20; configurations 2, 4a, 4b)  G consecutive steps are handed to the library in ONE call
      (alilqr_solve_batches): it solves them as one batch, the resident lane groups pull problems of all G batches
      from one queue.  --pipeline calls (default 2) are in flight on as many streams, so the tail of one call
      overlaps with the next.  Needs no more than HIP's default hardware queues.
  --group 1  one alilqr_solve_batch per step on --pipeline (default 12, at most 16) streams in turn; the process asks
      the HIP runtime for 16 hardware queues (--hw-queues): with the default 4, streams share queues and their
      kernels run one after the other.  (Configurations 3 and 5 take optional inputs and always run this way.)
All K steps complete inside the timed region; --group 1 --pipeline 1 gives the one-launch-at-a-time number.

  python bench.py --gpus N --steps K --warmup W [--config 2|3|4a|4b|5] [--scaling weak|strong] [--batch B]

--config selects the other BASELINE configurations (3: kinematic model + rectangle obstacles, C++ generation;
4a / 4b: dynamic model, H = 100, Python generations; 5: multi-start, 256 starts per scenario).  --scaling strong
shards a fixed global batch (--batch) over the ranks (whole scenarios for config 5); weak gives every rank --batch.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from ilqr_amd import batch, capi, problems, sharding  # noqa: E402

# Algorithmic fp64 operation counts per node, dense as the reference computes them (4-state / 6-state models);
# a transcendental call is counted as ONE operation.
FLOPS = {4: dict(bwd=469, lin=60 + 4, quad=40, dyn=30 + 6, cost=60),
         6: dict(bwd=1752, lin=180 + 4, quad=64, dyn=50 + 6, cost=96)}
FP64_VALU_PEAK_TFLOPS = 78.6     # MI355X vector fp64 (half the 157.3 TF fp32 vector rate)
HBM_PEAK_GBS = 8000.0
STARTS = 256                     # config 5: starts per scenario

CONFIGS = {
    "2": dict(batch=4096, scaling="weak", label="lat-bicycle AL-iLQR (ILQR + LatBicycleKinematicNode, box constraints), H=50"),
    "3": dict(batch=65536, scaling="strong", label="kinematic bicycle + linear table + rectangle obstacles (NewALILQR), H=50"),
    "4a": dict(batch=16384, scaling="strong", label="dynamic bicycle (ILQR + FullBicycleDynamicNode, box), H=100"),
    "4b": dict(batch=16384, scaling="strong", label="dynamic bicycle (FastILQR + FastBicycleNode, box + linear rows), H=100"),
    "5": dict(batch=STARTS * 1024, scaling="strong", label="multi-start lat-bicycle (NewALILQR line search), 256 starts x scenarios, H=50"),
}


def make_workload(config, B, seed):
    """(problem tuple, optional-input kwargs) for B problems of a configuration."""
    if config == "2":
        return problems.lat_legacy(B, 50, seed=seed), {}
    if config == "3":
        return problems.kin_rect(B, seed=1 + seed)
    if config == "4a":
        return problems.full_legacy("dyn", B, 100, 2 + seed), {}
    if config == "4b":
        return problems.fast_box_linear(B, seed=2 + seed), {}
    if config == "5":
        *prob, uo = problems.multi_start(B // STARTS, STARTS, seed=3 + seed)
        return tuple(prob), dict(u_offset=uo)
    raise SystemExit(f"unknown --config {config}")


def algorithmic_flops(status, H, n):
    n_inner = status[:, 0].astype(np.float64).sum()
    n_trials = status[:, 3].astype(np.float64).sum()
    f = FLOPS[n]
    return n_inner * H * (f["lin"] + f["quad"] + f["bwd"]) + n_trials * (H * f["dyn"] + (H + 1) * f["cost"])


def algorithmic_bytes(desc):
    B, H, n, m = desc.batch, desc.horizon, desc.n, desc.m
    return 8.0 * B * (n + (H + 1) * n + H * m + 1) + 16.0 * B


def kernel_source_id():
    """Hash of the solve kernel's sources: counters measured on another build are not reported as this build's."""
    h = hashlib.sha1()
    for f in ("solver.hpp", "models.hpp", "trig.hpp", "launch.hpp", "solve_family.hip"):
        with open(os.path.join(ROOT, "ilqr_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


def measured_counters(config):
    """HBM traffic per launch and VALU counters from the committed rocprofv3 --pmc passes of THIS build (the summary's
    kernel_source_id must be the tree's), or None.  profiles/r03_pmc_<config>.json, written by tools/profile_round.sh."""
    for name in (f"r03_pmc_{config}.json", "r02_pmc.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as fh:
                c = json.load(fh)
        except (OSError, ValueError):
            continue
        if c.get("kernel_source_id") == kernel_source_id() and str(c.get("config")) == config:
            c["source"] = "profiles/" + name
            return c
    return None


def cpu_baseline(config, sample, seed=0):
    """The CPU oracle (a restatement of the reference, 'port') on a bounded sample of the configuration's workload:
    its first `sample` problems (whole scenarios for the multi-start configuration), OpenMP over problems."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    threads = max(1, min(16, os.cpu_count() or 1, oracle_lib.lib().oracle_max_threads()))
    if config == "5":
        sample = max(STARTS, sample // STARTS * STARTS)
    prob, kw = make_workload(config, sample, seed)
    t0 = time.perf_counter()
    oracle_lib.solve_batch(*prob, threads=threads, extras=False, **kw)
    dt = time.perf_counter() - t0
    return dict(value=sample / dt, unit="solves/s", cores=threads, kind="port",
                sample=f"the first {sample} problems of rank 0's batch of configuration {config}, OpenMP over problems, {dt:.2f} s wall")


# problems of the CPU-baseline sample per configuration: about 3 - 20 s of work on 16 host cores
CPU_SAMPLE = {"2": 4096, "3": 2048, "4a": 1024, "4b": 2048, "5": 4096}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="2", choices=sorted(CONFIGS))
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"])
    ap.add_argument("--batch", type=int, default=0, help="problems per GPU (weak) or in total (strong); 0 = the configuration's")
    ap.add_argument("--cpu-sample", type=int, default=0, help="problems of the CPU-baseline sample (0 = the configuration's)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per problem (0 = the library's choice)")
    ap.add_argument("--max-waves", type=int, default=0,
                    help="resident waves per launch (0 = the library's choice); with fewer waves than problem pairs a lane "
                         "group pulls its next problem when it finishes one")
    ap.add_argument("--group", type=int, default=20,
                    help="steps (batches) per library call: 1 = one alilqr_solve_batch per step; > 1 = alilqr_solve_batches")
    ap.add_argument("--pipeline", type=int, default=0,
                    help="calls in flight (HIP streams used in turn); 0 = 2 with --group > 1, 12 with --group 1")
    ap.add_argument("--hw-queues", type=int, default=16,
                    help="GPU_MAX_HW_QUEUES for this process (HIP's default of 4 makes streams share a hardware "
                         "queue, and kernels on one queue run one after the other); 0 leaves the environment's value")
    ap.add_argument("--no-order-hint", action="store_true", help="skip the second timed region (pull-order hint)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: take the multi-rank path (process group, gather, barriers) with one rank too")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL) is the product path; gloo only rehearses the multi-rank control flow on a box "
                         "with fewer GPUs than ranks (ranks then share devices and the gather goes through the host)")
    args = ap.parse_args()
    if args.steps < 1:
        raise SystemExit("--steps must be >= 1")

    if args.hw_queues > 0:
        os.environ["GPU_MAX_HW_QUEUES"] = str(min(32, args.hw_queues))     # read by the HIP runtime when it starts
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device")
    if args.backend == "gloo":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    multi = world > 1 or args.force_dist
    if multi:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    if args.lanes or args.max_waves:
        capi.set_tuning(lanes=args.lanes, max_waves=args.max_waves)
    cfg = CONFIGS[args.config]
    scaling = args.scaling or cfg["scaling"]
    total = args.batch or cfg["batch"]
    unit = STARTS if args.config == "5" else 1          # config 5 shards whole scenarios
    if scaling == "strong":
        lo, hi = sharding.shard_range(total // unit, rank, world)
        B, global_B = (hi - lo) * unit, (total // unit) * unit
        prob, kw = make_workload(args.config, global_B, 0)     # every rank generates the global batch, takes its shard
        desc, x0, goals, uo = sharding.local_problem(prob[0], prob[1], prob[2], kw.get("u_offset"), lo * unit, hi * unit)
        prob = (desc, x0, goals) + tuple(prob[3:])
        if uo is not None:
            kw["u_offset"] = uo
    else:
        B = (total // unit) * unit
        global_B = B * world
        prob, kw = make_workload(args.config, B, rank)
    desc = prob[0]
    H, n = desc.horizon, desc.n
    if B < 1:
        raise SystemExit("this rank has no problems: use a larger --batch")

    group = max(1, args.group)
    if kw or args.config in ("3", "5"):
        group = 1                                   # optional inputs (rectangles, control offsets): one call per step
    depth = max(1, min(16, args.pipeline or (2 if group > 1 else 12)))
    streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
    if group > 1:
        # every batch of a call has start states of its own (seeded by rank and position in the call)
        x0s = [prob[1]] + [make_workload(args.config, B, 1000 * (rank + 1) + j)[0][1] for j in range(1, group)]
        if scaling == "strong":
            x0s = [prob[1]] * group                 # the sharded global batch, repeated
        solvers = [batch.MultiBatchSolver(desc, x0s, *prob[2:6], extras=False, device=dev) for _ in range(depth)]
    else:
        solvers = [batch.BatchSolver(*prob, extras=False, device=dev, **kw) for _ in range(depth)]
    single = solvers[0] if group == 1 else batch.BatchSolver(*prob, extras=False, device=dev, **kw)
    gdev = dev if args.backend == "nccl" else torch.device("cpu")
    # The only exchange of the job: one gather (RCCL) of the cost vectors the solve streams hold when the K steps
    # are done, inside the timed region.  The solves themselves need no collective.
    maxB = B
    if multi:
        t = torch.tensor([B], dtype=torch.int64, device=gdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        maxB = int(t.item())
    final = torch.zeros((depth, maxB), dtype=torch.float64, device=gdev) if multi else None
    gathered = torch.empty((world, depth, maxB), dtype=torch.float64, device=gdev) if multi else None

    def last_out(sv):
        return sv.out[0] if group > 1 else sv.out

    def gather_costs():
        for i, sv in enumerate(solvers):
            torch.cuda.current_stream(dev).wait_stream(streams[i])
            final[i, :B].copy_(last_out(sv)["cost"])
        dist.all_gather_into_tensor(gathered.view(-1), final.view(-1))

    orders = {}          # (stream slot, batches in the call) -> pull order from the previous solve of those batches

    def call(k, n, ev=None, hinted=False):
        """library call number k: n steps (batches) on stream k % depth"""
        i = k % depth
        with torch.cuda.stream(streams[i]):
            if ev is not None:
                ev[0].record(streams[i])
            if group > 1:
                solvers[i].run(streams[i].cuda_stream, n=n, order=orders.get((i, n)) if hinted else None)
            else:
                solvers[i].run(streams[i].cuda_stream)
            if ev is not None:
                ev[1].record(streams[i])
        return solvers[i]

    def calls_of(steps):
        """the steps of a region as library calls: (steps in the call) per call"""
        return [min(group, steps - s) for s in range(0, steps, group)]

    # set-up, not a step: every stream's first solve allocates that stream's scratch inside the library
    for i in range(depth):
        call(i, group)
    torch.cuda.synchronize(dev)

    for k, ns in enumerate(calls_of(args.warmup)):
        call(k, ns)
    if multi:
        gather_costs()                      # also brings the communicator up before the timed region
    torch.cuda.synchronize(dev)
    if multi:
        dist.barrier()
    torch.cuda.synchronize(dev)
    plan = calls_of(args.steps)

    def timed_region(hinted):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in plan]
        torch.cuda.synchronize(dev)
        if multi:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        sv = None
        for k, ns in enumerate(plan):
            sv = call(k, ns, ev[k], hinted)
        if multi:
            gather_costs()
        torch.cuda.synchronize(dev)
        if multi:
            dist.barrier()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        if multi:
            tmax = torch.tensor([dt], dtype=torch.float64, device=gdev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt, ev, sv

    dt, ev, sv = timed_region(False)
    # Pull-order hint (group mode), reported next to the main number and never as `value`.  A planning loop solves,
    # cycle after cycle, problems that differ little from the previous cycle's; the inner-iteration counts of the
    # previous cycle predict this cycle's.  Emulated here: every call's batches are solved once OUTSIDE the timed
    # region from start states moved by 1 % of their spread ("the previous cycle"), and the K steps are timed once more
    # with every call pulling its problems longest-first by those counts (alilqr_extras::order passed to alilqr_solve_batches).
    hint = None
    if group > 1 and not args.no_order_hint:
        spread = np.array([0.0, 1.0, 0.2, 0.05]) if n == 4 else np.array([0.0, 1.0, 0.1, 0.0, 1.0, 0.0])
        rng = np.random.default_rng(12345 + rank)
        moved = float(os.environ.get("ALILQR_BENCH_HINT_SPREAD", "0.01"))     # development: 0 = a perfect hint
        prev_x0s = [np.asarray(x) + moved * spread * rng.standard_normal(np.asarray(x).shape) for x in x0s]
        prev = batch.MultiBatchSolver(desc, prev_x0s, *prob[2:6], extras=False, device=dev)
        prev.run()
        torch.cuda.synchronize(dev)
        for ns in set(plan):
            od = prev.order_by_last_solve(ns)
            for i in range(depth):
                orders[(i, ns)] = od
        prev_iters = torch.cat([o["status"][:, 0] for o in prev.out]).cpu().numpy()
        del prev
        dt_h, _, sv_h = timed_region(True)
        now_iters = torch.cat([o["status"][:, 0] for o in sv_h.out]).cpu().numpy()
        ra, rb = np.argsort(np.argsort(prev_iters)), np.argsort(np.argsort(now_iters))
        hint = {"value": global_B * args.steps / dt_h, "ms_per_step": dt_h / args.steps * 1e3,
                "source": "inner-iteration counts of the same batches solved, outside the timed region, from start states "
                          "moved by 1 % of their spread (the previous cycle of a planning loop); longest first",
                "rank_correlation_with_actual_counts": float(np.corrcoef(ra, rb)[0, 1])}
    # per-problem status words of the first call's batches (the algorithmic operation count comes from them); a later,
    # smaller call on the same stream slot re-solves a prefix of the same batches and leaves the same words
    last_n = plan[0]
    outs = solvers[0].out[:last_n] if group > 1 else [solvers[0].out]
    out = outs[0]

    # outside the timed region: the same launch alone on the device, for the record
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(streams[0]):
        single.run(streams[0].cuda_stream)                     # set-up: scratch of this size
        e0.record(streams[0]); single.run(streams[0].cuda_stream); e1.record(streams[0])
    torch.cuda.synchronize(dev)
    lone_ms = e0.elapsed_time(e1)
    # and a launch that leaves every wave a SIMD to itself (1024 problems = 512 waves on 1024 SIMDs): the latency of
    # one iLQR iteration when nothing else competes for the wave's SIMD
    light_B = min(B, 1024)
    lprob, lkw = make_workload(args.config, light_B, rank)
    light = batch.BatchSolver(*lprob, extras=False, device=dev, **lkw)
    with torch.cuda.stream(streams[0]):
        light.run(streams[0].cuda_stream)                      # set-up: scratch of this size
        e0.record(streams[0]); lout = light.run(streams[0].cuda_stream); e1.record(streams[0])
    torch.cuda.synchronize(dev)
    light_ms = e0.elapsed_time(e1)
    light_iters = int(lout["status"][:, 0].max().item())

    # a launch = one library call = `steps_per_launch` steps; duration of the first call's size class only
    full = [i for i, n_ in enumerate(plan) if n_ == last_n]
    kernel_ms = float(np.mean([ev[i][0].elapsed_time(ev[i][1]) for i in full]))
    status = np.concatenate([o["status"].cpu().numpy() for o in outs])
    cost = np.concatenate([o["cost"].cpu().numpy() for o in outs])
    flops = algorithmic_flops(status, H, n)                      # of one launch (last_n steps)
    tf = flops / (kernel_ms * 1e-3) / 1e12                       # per launch, as each launch experienced it
    tf_job = flops / last_n * args.steps / dt / 1e12             # all steps of the timed region over its wall time
    single_status = single.out["status"].cpu().numpy()
    hit_cap = float(np.mean((status[:, 2] & capi.FLAG_OUTER_CAP) != 0))
    viol_ok = float(np.mean((status[:, 2] & capi.FLAG_VIOL_OK) != 0))

    if rank == 0:
        value = global_B * args.steps / dt
        pmc = measured_counters(args.config)
        line = {
            "metric": f"converged AL-iLQR solves/sec (H={H}, batch N)",
            "value": value,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"BASELINE config {args.config}: {cfg['label']}, "
                                   + (f"batch {B} per GPU, start states seeded by rank" if scaling == "weak"
                                      else f"global batch {global_B} sharded over {world} GPU(s)"),
                       "config": args.config, "batch_per_gpu": B, "global_batch": global_B, "horizon": H,
                       "lanes_per_problem": args.lanes or "auto",
                       "steps_per_call": group, "calls_in_flight": depth,
                       "entry_point": "alilqr_solve_batches" if group > 1 else "alilqr_solve_batch",
                       "with_order_hint": hint,
                       "lone_launch_ms": lone_ms,
                       # latency of one iLQR iteration (linearise, backward sweep, line search) of one problem with the
                       # whole batch resident: a lone launch lasts as long as its longest solve
                       "per_iter_latency_us": lone_ms * 1e3 / max(1, int(single_status[:, 0].max())),
                       # the same with 1024 problems resident (one wave per SIMD at most)
                       "light_load": {"batch": light_B, "launch_ms": light_ms,
                                      "per_iter_latency_us": light_ms * 1e3 / max(1, light_iters)},
                       "mean_inner_iters": float(status[:, 0].mean()), "mean_rollouts": float(status[:, 3].mean()),
                       # "converged" = reached the reference generation's own stopping state: violation below its
                       # tolerance, or its outer-iteration cap (the reference stops there too); both shares are given
                       "frac_viol_below_tol": viol_ok, "frac_outer_cap": hit_cap,
                       "solves_per_s_viol_below_tol": value * viol_ok,
                       "frac_nonfinite": float(np.mean(~np.isfinite(cost)))},
            "roofline": {"bound": "valu_fp64", "achieved": tf, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tf / FP64_VALU_PEAK_TFLOPS,
                         # HBM bytes per launch from the PMC passes of this build (null when none is committed for it)
                         # (measured per solve on launches of the configuration's batch, times the problems of this launch)
                         "traffic": (pmc["hbm_bytes_per_solve"] * B * last_n) if pmc and pmc.get("hbm_bytes_per_solve") else
                                    (pmc.get("hbm_bytes_per_launch") if pmc else None),
                         "counters": pmc,
                         # kernel_ms is a launch as it experienced the machine: `launches_in_flight` launches share the GPU, so
                         # a launch lasts about that many times its share; achieved_job / frac_job are the algorithmic
                         # flops of all K launches over the wall time of the timed region
                         "kernel": "alilqr::solve_kernel", "kernel_ms": kernel_ms, "launches_in_flight": min(depth, len(plan)),
                         "steps_per_launch": last_n,
                         "achieved_job": tf_job, "frac_job": tf_job / FP64_VALU_PEAK_TFLOPS,
                         "algorithmic_flops_per_launch": flops,
                         "algorithmic_hbm_bytes_per_launch": algorithmic_bytes(desc) * last_n,
                         "hbm_GBs_algorithmic": algorithmic_bytes(desc) * last_n / (kernel_ms * 1e-3) / 1e9,
                         "hbm_peak_GBs": HBM_PEAK_GBS, "kernel_source_id": kernel_source_id()},
        }
        if world == 1 and not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline(args.config, min(B, args.cpu_sample or CPU_SAMPLE[args.config]))
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
