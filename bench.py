#!/usr/bin/env python3
"""bench.py -- Gauss-Newton iterations/s of the pose-graph hot path on MI355X.

A "step" is ONE Gauss-Newton iteration of `PoseGraph::optimize`'s loop body (reference
src/mapping/pose_graph_optimization.rs:269-301): linearise + assemble, factor, solve, update, chi2,
|dx| -- all on the GPU, graph state resident in HBM, no host round trip inside the timed region
(the convergence break is disabled so that exactly K steps run; after convergence a step does the
same work on the same pattern).  Default workload = BASELINE.json configs[1]: intel.g2o, fp64.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload intel|m3500|dlr|grid:WxH[:E]] [--precision f64|f32]

N > 1 (launched by torch.distributed.run, one rank per GPU): each rank optimises its own replica of
the workload -- independent graphs, no data-path collective ("replicas", weak scaling); the
barrier and the max-over-ranks time use RCCL through torch.distributed.

Prints ONE JSON line (rank 0) with the contract keys plus `roofline` (dominant kernel class, HIP
event timing on the library's own stream) and `cpu_baseline` (the CPU oracle, 1 thread, bounded
sample, rank 0 at N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s measured copy)

WORKLOADS = {"intel": "intel", "m3500": "input_M3500_g2o", "dlr": "dlr", "pose-pose": "simulation-pose-pose",
             "pose-landmark": "simulation-pose-landmark"}


def g2o_file(name):
    return os.path.join(ROOT, "tests", "golden", "g2o", WORKLOADS[name] + ".g2o")


def dist_env():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def timed_steps(run_steps, sync, barrier, all_max, steps, warmup, reset=None):
    """The measurement contract: W untimed steps, barrier + sync, EXACTLY K steps, sync + barrier,
    MAX over ranks.  Kept free of GPU specifics so the gloo CPU test can drive it."""
    if warmup > 0:
        run_steps(warmup)
    sync()
    if reset is not None:
        reset()
    sync()
    barrier()
    t0 = time.perf_counter()
    run_steps(steps)
    sync()
    barrier()
    dt = time.perf_counter() - t0
    return all_max(dt)


def make_graph(workload, precision, device):
    from rustrobotics_amd import PoseGraph, PoseGraphSolver
    if workload.startswith("grid:"):
        parts = workload.split(":")
        w, h = (int(x) for x in parts[1].lower().split("x"))
        e = int(parts[2]) if len(parts) > 2 else 0
        return PoseGraph.synthetic_grid(w, h, e, solver=PoseGraphSolver.GaussNewton, precision=precision, device=device)
    return PoseGraph.new(g2o_file(workload), PoseGraphSolver.GaussNewton, precision=precision, device=device)


def cpu_baseline(workload, budget_s=12.0):
    """The CPU oracle (scalar fp64 restatement of the reference loop, ordering + symbolic + numeric
    factorisation redone every iteration like the reference's UMFPACK path), 1 thread."""
    from oracle.oracle import OracleGraph
    if workload.startswith("grid:"):
        from rustrobotics_amd import synthetic_grid_arrays
        parts = workload.split(":")
        w, h = (int(x) for x in parts[1].lower().split("x"))
        arrays = synthetic_grid_arrays(w, h, int(parts[2]) if len(parts) > 2 else 0)
        load = lambda: OracleGraph.from_arrays(*arrays)  # noqa: E731
    else:
        load = lambda: OracleGraph.load(g2o_file(workload))  # noqa: E731
    iters, spent, restarts, last = 0, 0.0, 0, None
    while spent < budget_s:
        g = load()
        t0 = time.perf_counter()
        errs = g.optimize(10)
        spent += time.perf_counter() - t0
        iters += len(errs) - 1
        restarts += 1
        last = errs
    return {"value": iters / spent, "unit": "GN iterations/s", "cores": 1, "kind": "port",
            "sample": f"{workload}: {iters} GN iterations in {restarts} runs of optimize(10) "
                      f"(stops at |dx|<1e-4), {spent:.1f} s of CPU",
            "chi2_final": float(last[-1]), "errors": [float(x) for x in last]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="intel")
    ap.add_argument("--precision", default="f64", choices=["f64", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    rank, local_rank, world = dist_env()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    g = make_graph(args.workload, args.precision, local_rank)
    state0 = g.state()

    def run_steps(k):
        g.iterate_async(k)

    def sync():
        g.sync()                    # the library's own HIP stream
        torch.cuda.synchronize()

    def barrier():
        if use_dist:
            dist.barrier()

    def all_max(x):
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    dt = timed_steps(run_steps, sync, barrier, all_max, args.steps, args.warmup, reset=lambda: g.set_state(state0))
    stats = g.stats()
    total_steps = args.steps * world
    value = total_steps / dt

    out = None
    if rank == 0:
        # correctness leg: the reference's bench shape, optimize(10) from the initial state
        g.set_state(state0)
        t0 = time.perf_counter()
        errors = g.optimize(10)
        opt_ms = (time.perf_counter() - t0) * 1e3
        # per-kernel-class timing with HIP events on the library's stream (eager launches)
        g.set_state(state0)
        prof = g.profile(20)
        per_iter_us = {k: 1e3 * v[0] / 20 for k, v in prof.items()}
        class_bytes = {"linearize": stats["bytes_linearize"], "factor": stats["bytes_factor"],
                       "solve": stats["bytes_solve"], "update": stats["bytes_update"]}
        dom = max(class_bytes, key=lambda k: per_iter_us.get(k, 0.0) + (per_iter_us.get("bigfront", 0.0) if k == "factor" else 0.0))
        dom_us = per_iter_us[dom] + (per_iter_us.get("bigfront", 0.0) if dom == "factor" else 0.0)
        n_launch = prof[dom][1] / 20 + (prof["bigfront"][1] / 20 if dom == "factor" else 0)
        achieved = class_bytes[dom] / (dom_us * 1e-6) / 1e9 if dom_us > 0 else 0.0
        roofline = {"bound": "hbm", "kernel": {"linearize": "k_linearize", "factor": "k_factor_tasks",
                                               "solve": "k_solve_tasks", "update": "k_update"}[dom],
                    "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                    "traffic": None, "launches_per_step": n_launch,
                    "avg_launch_us": dom_us / max(n_launch, 1), "algorithmic_bytes_per_launch": class_bytes[dom] / max(n_launch, 1),
                    "per_step_us_by_kernel_class": per_iter_us}
        out = {
            "metric": "GN iterations/s", "value": value, "unit": "GN iterations/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "reference dataset file" if not args.workload.startswith("grid:") else "synthetic",
            "config": {"workload": f"{args.workload} ({g.num_nodes} poses / {g.num_edges} edges / dim {g.len}), "
                                   f"Gauss-Newton, one independent replica per GPU", "solver": "GaussNewton",
                       "parallelism": "replicas" if world > 1 else "single"},
            "edges_iters_per_s": value * g.num_edges,
            "optimize10_ms": opt_ms, "errors": [float(e) for e in errors],
            "analyze_ms": stats["analyze_ms"], "parse_ms": stats["parse_ms"],
            "launches_per_step": stats["n_launches_per_iter"], "supernodes": stats["n_supernodes"],
            "factor_flops": stats["factor_flops"], "algorithmic_bytes_per_step": sum(class_bytes.values()),
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(args.workload)
            ref = np.array(cb.pop("errors"))
            out["cpu_baseline"] = cb
            out["chi2_final"] = float(errors[-1])
            out["chi2_rel_diff_vs_cpu"] = abs(errors[-1] - ref[-1]) / ref[-1]
            out["speedup_vs_cpu_baseline"] = value / cb["value"]
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
