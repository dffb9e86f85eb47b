#!/usr/bin/env python3
"""Throughput of the batched AL-iLQR solve path on MI355X.

One "step" = one complete solve of the workload batch: 4096 lateral-bicycle
problems (H=50, box-constrained, start states randomised with seed = rank), the
configuration BASELINE.json quotes the solves/sec metric on.  Inputs are
resident in HBM before the timed region; every rank owns its own batches (weak
scaling, no collective inside the solve; one gather of the final cost vectors
over RCCL at the end of the K steps is the only exchange).

Consecutive steps are independent batches.  They are enqueued on --pipeline (default 6) HIP streams in turn, each
with its own output buffers and its own scratch inside the library, so the next batches fill the SIMDs that the
tail of a batch leaves idle (a batch ends when its longest solve ends; the mean wave is busy ~40 % of a lone
launch).  The process asks the HIP runtime for 16 hardware queues (--hw-queues): with the default 4, streams share
queues and their kernels run one after the other.  All K steps complete inside the timed region; --pipeline 1
gives the one-launch-at-a-time number.

  python bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from ilqr_amd import batch, capi, problems  # noqa: E402

# Algorithmic fp64 operation counts per node (lateral model), dense as the
# reference computes them; a transcendental call is counted as ONE operation.
LAT_FLOPS = dict(bwd=469, lin=60 + 4, quad=40, dyn=30 + 6, cost=60)
FP64_VALU_PEAK_TFLOPS = 78.6     # MI355X vector fp64 (half the 157.3 TF fp32 vector rate)
HBM_PEAK_GBS = 8000.0


def algorithmic_flops(status, H):
    n_inner = status[:, 0].astype(np.float64).sum()
    n_trials = status[:, 3].astype(np.float64).sum()
    f = LAT_FLOPS
    return n_inner * H * (f["lin"] + f["quad"] + f["bwd"]) + n_trials * (H * f["dyn"] + (H + 1) * f["cost"])


def algorithmic_bytes(desc):
    B, H, n, m = desc.batch, desc.horizon, desc.n, desc.m
    return 8.0 * B * (n + (H + 1) * n + H * m + 1) + 16.0 * B


def cpu_baseline(sample, seconds_hint=20.0):
    """The CPU oracle (a restatement of the reference, 'port') on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    threads = max(1, min(16, os.cpu_count() or 1, oracle_lib.lib().oracle_max_threads()))
    desc, x0, goals, Q, R, cons = problems.lat_legacy(sample, 50, seed=0)
    t0 = time.perf_counter()
    oracle_lib.solve_batch(desc, x0, goals, Q, R, cons, threads=threads, extras=False)
    dt = time.perf_counter() - t0
    return dict(value=sample / dt, unit="solves/s", cores=threads, kind="port",
                sample=f"the first {sample} problems of the 4096-problem batch of rank 0, OpenMP over problems, {dt:.2f} s wall")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--cpu-sample", type=int, default=4096)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per problem (0 = the library's choice)")
    ap.add_argument("--pipeline", type=int, default=6, help="batches in flight (HIP streams used in turn)")
    ap.add_argument("--hw-queues", type=int, default=16,
                    help="GPU_MAX_HW_QUEUES for this process (HIP's default of 4 makes streams share a hardware "
                         "queue, and kernels on one queue run one after the other); 0 leaves the environment's value")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: take the multi-rank path (process group, per-step gather, barriers) with one rank too")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL) is the product path; gloo only rehearses the multi-rank control flow on a box "
                         "with fewer GPUs than ranks (ranks then share devices and the gather goes through the host)")
    args = ap.parse_args()

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

    if args.lanes:
        capi.set_tuning(lanes=args.lanes)
    B, H = args.batch, 50
    desc, x0, goals, Q, R, cons = problems.lat_legacy(B, H, seed=rank)
    depth = max(1, min(8, args.pipeline))
    solvers = [batch.BatchSolver(desc, x0, goals, Q, R, cons, extras=False, device=dev) for _ in range(depth)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
    gdev = dev if args.backend == "nccl" else torch.device("cpu")
    # The only exchange of the job: one gather (RCCL) of the cost vectors the solve streams hold when the K steps
    # are done, inside the timed region.  The solves themselves need no collective.
    final = torch.empty((depth, B), dtype=torch.float64, device=gdev) if multi else None
    gathered = torch.empty((world, depth, B), dtype=torch.float64, device=gdev) if multi else None

    def gather_costs():
        for i, sv in enumerate(solvers):
            torch.cuda.current_stream(dev).wait_stream(streams[i])
            final[i].copy_(sv.out["cost"])
        dist.all_gather_into_tensor(gathered.view(-1), final.view(-1))

    # set-up, not a step: every stream's first solve allocates that stream's scratch inside the library
    for sv, st in zip(solvers, streams):
        with torch.cuda.stream(st):
            sv.run(st.cuda_stream)
    torch.cuda.synchronize(dev)

    def step(k, ev=None):
        i = k % depth
        with torch.cuda.stream(streams[i]):
            if ev is not None:
                ev[0].record(streams[i])
            out = solvers[i].run(streams[i].cuda_stream)
            if ev is not None:
                ev[1].record(streams[i])
        return out

    for k in range(args.warmup):
        step(k)
    if multi:
        gather_costs()                      # also brings the communicator up before the timed region
    torch.cuda.synchronize(dev)
    if multi:
        dist.barrier()
    torch.cuda.synchronize(dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        out = step(k, ev[k])
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

    # outside the timed region: the same launch alone on the device, for the record
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(streams[0]):
        e0.record(streams[0]); solvers[0].run(streams[0].cuda_stream); e1.record(streams[0])
    torch.cuda.synchronize(dev)
    lone_ms = e0.elapsed_time(e1)

    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    status = out["status"].cpu().numpy()
    cost = out["cost"].cpu().numpy()
    flops = algorithmic_flops(status, H)
    tf = flops / (kernel_ms * 1e-3) / 1e12                       # per launch, as each launch experienced it
    tf_job = flops * args.steps / dt / 1e12                      # all launches of the timed region over its wall time
    hit_cap = float(np.mean((status[:, 2] & capi.FLAG_OUTER_CAP) != 0))

    if rank == 0:
        # counters of the same launch, collected in separate rocprofv3 --pmc passes (tools/profile_round.sh)
        traffic = valu_active = executed = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc.json")
        if os.path.exists(pmc):
            try:
                with open(pmc) as fh:
                    counters = json.load(fh)
                traffic = counters.get("hbm_bytes_per_launch")
                if counters.get("SQ_WAVE_CYCLES"):
                    valu_active = counters["SQ_ACTIVE_INST_VALU"] / counters["SQ_WAVE_CYCLES"]
                executed = counters.get("executed_fp64_lane_flops_per_launch")
            except (OSError, ValueError, KeyError):
                traffic = valu_active = executed = None
        line = {
            "metric": "converged AL-iLQR solves/sec (H=50, batch N)",
            "value": world * B * args.steps / dt,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "lat-bicycle AL-iLQR (ILQR + LatBicycleKinematicNode, box constraints), H=50, "
                                   f"batch {B} per GPU, start states x0=(0,U(-1,1),U(-0.2,0.2),U(-0.05,0.05)) seed=rank",
                       "batch_per_gpu": B, "horizon": H, "mode": "legacy", "model": "lat",
                       "lanes_per_problem": args.lanes or "auto",
                       "batches_in_flight": depth, "lone_launch_ms": lone_ms,
                       # latency of one iLQR iteration (linearise, backward sweep, line search) of one problem with the
                       # whole batch resident: a lone launch lasts as long as its longest solve
                       "per_iter_latency_us": lone_ms * 1e3 / max(1, int(status[:, 0].max())),
                       "mean_inner_iters": float(status[:, 0].mean()), "mean_rollouts": float(status[:, 3].mean()),
                       "frac_outer_cap": hit_cap, "frac_nonfinite": float(np.mean(~np.isfinite(cost)))},
            "roofline": {"bound": "valu_fp64", "achieved": tf, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tf / FP64_VALU_PEAK_TFLOPS, "traffic": traffic,
                         "valu_active_share_of_wave_cycles": valu_active,
                         # fp64 flops the hardware executed (FMA, ADD, MUL instruction counters x 64 lanes, masked lanes
                         # included) over the job's wall time: the machine-side utilisation beside the algorithmic one
                         "executed_fp64_TFLOPs_job": (executed * args.steps / dt / 1e12) if executed else None,
                         "executed_fp64_frac_job": (executed * args.steps / dt / 1e12 / FP64_VALU_PEAK_TFLOPS) if executed else None,
                         "kernel": "alilqr::solve_kernel", "kernel_ms": kernel_ms,
                         "achieved_job": tf_job, "frac_job": tf_job / FP64_VALU_PEAK_TFLOPS,
                         "algorithmic_flops_per_launch": flops,
                         "algorithmic_hbm_bytes_per_launch": algorithmic_bytes(desc),
                         "hbm_GBs_algorithmic": algorithmic_bytes(desc) / (kernel_ms * 1e-3) / 1e9,
                         "hbm_peak_GBs": HBM_PEAK_GBS},
        }
        if world == 1 and not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
