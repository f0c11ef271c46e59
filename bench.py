#!/usr/bin/env python3
"""bench.py -- Gauss-Newton iterations/s of the pose-graph hot path on MI355X.

A "step" is ONE Gauss-Newton iteration of `PoseGraph::optimize`'s loop body (reference
src/mapping/pose_graph_optimization.rs:269-301): linearise + assemble, factor, solve, update, chi2,
|dx| -- all on the GPU, graph state resident in HBM, no host round trip inside the timed region
(the convergence break is disabled so that exactly K steps run; after convergence a step does the
same work on the same pattern).  Default workload = BASELINE.json configs[1]: intel.g2o, fp64.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload intel|m3500|dlr|sphere2500|torus3d|parking-garage|grid:WxH[:E]] [--precision f64|f32|mixed] [--shard]

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
# MI355X_MICROARCH.md: f32-input MFMA 157.3 TFLOP/s dense; f64 matrix/vector 78.6 TFLOP/s (datasheet)
MFMA_PEAK_TFLOPS = {"f32": 157.3, "mixed": 157.3, "f64": 78.6}

WORKLOADS = {"intel": "intel", "m3500": "input_M3500_g2o", "dlr": "dlr", "pose-pose": "simulation-pose-pose",
             "pose-landmark": "simulation-pose-landmark", "sphere2500": "sphere2500",
             "torus3d": "torus3D", "parking-garage": "parking-garage"}


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


def make_sharded_graph(workload, precision, device, rank, world):
    from rustrobotics_amd import PoseGraph, synthetic_grid_arrays
    if workload.startswith("grid:"):
        parts = workload.split(":")
        w, h = (int(x) for x in parts[1].lower().split("x"))
        arrays = synthetic_grid_arrays(w, h, int(parts[2]) if len(parts) > 2 else 0)
    else:
        arrays = PoseGraph.new(g2o_file(workload), precision=precision, device=device).graph_arrays()
    return PoseGraph.from_arrays(*arrays, precision=precision, device=device, rank=rank, world_size=world)


def cpu_baseline(workload, budget_s=12.0):
    """The CPU oracle (scalar fp64 restatement of the reference loop, ordering + symbolic + numeric
    factorisation redone every iteration like the reference's UMFPACK path), 1 thread.

    A lattice too large for the oracle to finish an iteration in the budget (the 1M-edge config
    needs minutes per iteration on one core) is sampled by a 100 x 100 lattice of the SAME generator
    and reported in edges*iterations/s, the size-independent half of BASELINE.json's metric."""
    from oracle.oracle import OracleGraph
    sample_note, n_edges = workload, None
    if workload.startswith("grid:"):
        from rustrobotics_amd import synthetic_grid_arrays
        parts = workload.split(":")
        w, h = (int(x) for x in parts[1].lower().split("x"))
        e = int(parts[2]) if len(parts) > 2 else 0
        if w * h > 12000:
            w, h, e = 100, 100, 0
            sample_note = f"100x100 lattice of the same generator (stand-in for {workload})"
        arrays = synthetic_grid_arrays(w, h, e)
        n_edges = len(arrays[2])
        load = lambda: OracleGraph.from_arrays(*arrays)  # noqa: E731
    else:
        load = lambda: OracleGraph.load(g2o_file(workload))  # noqa: E731
    iters, spent, restarts, last = 0, 0.0, 0, None
    while spent < budget_s:
        g = load()
        n_edges = g.num_edges
        t0 = time.perf_counter()
        errs = g.optimize(10)
        spent += time.perf_counter() - t0
        iters += len(errs) - 1
        restarts += 1
        last = errs
    out = {"value": iters / spent, "unit": "GN iterations/s", "cores": 1, "kind": "port",
           "sample": f"{sample_note}: {iters} GN iterations in {restarts} runs of optimize(10) "
                     f"(stops at |dx|<1e-4), {spent:.1f} s of CPU",
           "edges_iters_per_s": iters * n_edges / spent,
           "chi2_final": float(last[-1]), "errors": [float(x) for x in last]}
    if sample_note != workload:
        out["value"], out["unit"] = out["edges_iters_per_s"], "edges*iterations/s"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="intel")
    ap.add_argument("--precision", default="f64", choices=["f64", "f32", "mixed"],
                    help="f64 (reference arithmetic), f32, or mixed = f64 state/linearisation + f32 factor/solve")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shard", action="store_true",
                    help="N > 1 only: shard ONE graph over the ranks (own subtrees + shared top separators, two RCCL "
                         "all-reduces per iteration) instead of one replica per rank; strong scaling")
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

    shard = args.shard and use_dist
    if shard:
        g = make_sharded_graph(args.workload, args.precision, local_rank, rank, world)
        xbuf = []
        for which in (0, 1):
            _, n, es = g.exchange_info(which)
            t = torch.zeros(max(n, 1), dtype=torch.float64 if es == 8 else torch.float32, device="cuda")  # element size from the handle
            g.bind_exchange(which, t.data_ptr(), t.numel())
            xbuf.append(t)
    else:
        g = make_graph(args.workload, args.precision, local_rank)
    state0 = g.state()

    def run_steps(k):
        if not shard:
            g.iterate_async(k)
            return
        for _ in range(k):          # one GN iteration = 3 stages, 2 sum all-reduces over RCCL
            for stage in (0, 1):
                g.stage(stage)
                g.sync()
                dist.all_reduce(xbuf[stage], op=dist.ReduceOp.SUM)
                torch.cuda.current_stream().synchronize()
            g.stage(2)

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
    total_steps = args.steps * (1 if shard else world)   # sharded: all ranks work on the SAME K iterations
    value = total_steps / dt

    out = None
    if rank == 0 and shard:
        out = {"metric": "GN iterations/s", "value": value, "unit": "GN iterations/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": args.precision, "data": "synthetic" if args.workload.startswith("grid:") else "reference dataset file",
               "config": {"workload": f"{args.workload} ({g.num_nodes} poses / {g.num_edges} edges / dim {g.len}), Gauss-Newton, "
                                      f"ONE graph sharded over {world} ranks", "solver": "GaussNewton", "parallelism": "sharded%d" % world},
               "edges_iters_per_s": value * g.num_edges, "chi2_final": g.global_error(),
               "exchange_bytes_per_step": [int(t.numel() * t.element_size()) for t in xbuf],
               "roofline": None, "cpu_baseline": None}
    elif rank == 0:
        # correctness leg: the reference's bench shape, optimize(10) from the initial state
        g.set_state(state0)
        t0 = time.perf_counter()
        errors = g.optimize(10)
        opt_ms = (time.perf_counter() - t0) * 1e3
        # the reference's criterion closure (benches/graph_slam.rs:9-10): PoseGraph::new(file)?.optimize(10, false, false),
        # parsing, analysis, device set-up and tear-down all inside; only for file workloads
        closure_ms = None
        if not args.workload.startswith("grid:"):
            reps = []
            for _ in range(5):
                t0 = time.perf_counter()
                gg = make_graph(args.workload, args.precision, local_rank)
                gg.optimize(10)
                del gg
                reps.append((time.perf_counter() - t0) * 1e3)
            closure_ms = sorted(reps)[len(reps) // 2]
        # per-kernel-class timing with HIP events on the library's stream (eager launches)
        g.set_state(state0)
        prof = g.profile(20)
        per_iter_us = {k: 1e3 * v[0] / 20 for k, v in prof.items()}   # HIP events on the library's stream
        class_bytes = {"linearize": stats["bytes_linearize"], "factor": stats["bytes_factor"],
                       "solve": stats["bytes_solve"], "update": stats["bytes_update"]}
        kernel_of = {"linearize": "k_linearize", "factor": "k_factor_tasks", "solve": "k_solve_tasks",
                     "update": "k_update", "reduce": "k_finalize_slot", "big_assembly": "k_big_zero+k_big_assemble+k_big_extend_add",
                     "big_panel": "k_big_diag32+k_big_panel32", "big_update": "k_big_update", "mid_factor": "k_factor_mid",
                     "big_solve": "k_big_gemv_partial+k_big_gemv_finish+k_solve_mid"}
        dom = max((k for k in per_iter_us if k != "reduce"), key=lambda k: per_iter_us[k])
        n_launch = prof[dom][1] / 20
        dom_us = per_iter_us[dom]
        if dom == "big_update":
            # the rank updates of the huge fronts are dense contractions on the matrix cores
            achieved = stats["big_update_flops"] / (dom_us * 1e-6) / 1e12
            peak = MFMA_PEAK_TFLOPS[args.precision]
            roofline = {"bound": "mfma", "kernel": kernel_of[dom], "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                        "frac": achieved / peak, "traffic": None, "launches_per_step": n_launch,
                        "avg_launch_us": dom_us / max(n_launch, 1),
                        "algorithmic_flops_per_launch": stats["big_update_flops"] / max(n_launch, 1),
                        "per_step_us_by_kernel_class": per_iter_us}
        else:
            # fronts beyond LDS share the factor/solve byte budget with the LDS fronts
            nbytes = class_bytes.get(dom, stats["bytes_solve"] if dom == "big_solve" else stats["bytes_factor"])
            achieved = nbytes / (dom_us * 1e-6) / 1e9 if dom_us > 0 else 0.0
            roofline = {"bound": "hbm", "kernel": kernel_of[dom], "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "launches_per_step": n_launch,
                        "avg_launch_us": dom_us / max(n_launch, 1),
                        "algorithmic_bytes_per_launch": nbytes / max(n_launch, 1),
                        "per_step_us_by_kernel_class": per_iter_us}
        # graphs with fronts beyond LDS: the dense trailing update is the one kernel bounded by the matrix
        # cores; its utilisation is reported beside the dominant class whichever that is
        mfma_kernel = None
        if per_iter_us.get("big_update", 0) > 0:
            tf = stats["big_update_flops"] / (per_iter_us["big_update"] * 1e-6) / 1e12
            mfma_kernel = {"kernel": "k_big_update", "achieved": tf, "peak": MFMA_PEAK_TFLOPS[args.precision], "unit": "TFLOP/s",
                           "frac": tf / MFMA_PEAK_TFLOPS[args.precision], "us_per_step": per_iter_us["big_update"],
                           "launches_per_step": prof["big_update"][1] / 20, "flops_per_step": stats["big_update_flops"]}
        # HBM traffic of the dominant kernel: measured offline with rocprofv3 --pmc (separate passes),
        # kept in profiles/pmc_traffic.json; null when no measurement exists for this workload + kernel
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            ent = pmc.get(f"{args.workload}:{args.precision}")
            if not (ent and ent["kernel"] == roofline["kernel"]):   # the dominant class may be the MFMA kernel
                ent = pmc.get(f"{args.workload}:{args.precision}:{roofline['kernel']}")
            if ent and ent["kernel"] == roofline["kernel"]:
                roofline["traffic"] = ent["traffic_bytes_per_launch"]
                roofline["traffic_source"] = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH doubled per MI355X_MICROARCH.md)"
            # the dense trailing update against its OTHER roof: K = 128 columns per pass over the trailing matrix
            # bounds its arithmetic intensity, so the measured HBM traffic per launch prices it against HBM too
            ent = pmc.get(f"{args.workload}:{args.precision}:k_big_update")
            if ent and mfma_kernel:
                us = mfma_kernel["us_per_step"] / mfma_kernel["launches_per_step"]
                mfma_kernel["traffic"] = ent["traffic_bytes_per_launch"]
                mfma_kernel["hbm_achieved_GBps"] = ent["traffic_bytes_per_launch"] / (us * 1e-6) / 1e9
                mfma_kernel["hbm_frac"] = mfma_kernel["hbm_achieved_GBps"] / HBM_PEAK_GBPS
                mfma_kernel["flop_per_byte"] = mfma_kernel["flops_per_step"] / mfma_kernel["launches_per_step"] / ent["traffic_bytes_per_launch"]
        except (OSError, ValueError):
            pass
        out = {
            "metric": "GN iterations/s", "value": value, "unit": "GN iterations/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "strong" if shard else "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "reference dataset file" if not args.workload.startswith("grid:") else "synthetic",
            "config": {"workload": f"{args.workload} ({g.num_nodes} poses / {g.num_edges} edges / dim {g.len}), "
                                   f"Gauss-Newton, " + ("ONE graph sharded over the ranks" if shard else "one independent replica per GPU"),
                       "solver": "GaussNewton",
                       "parallelism": ("sharded%d" % world) if shard else ("replicas" if world > 1 else "single")},
            "edges_iters_per_s": value * g.num_edges,
            "optimize10_ms": opt_ms, "new_plus_optimize10_ms": closure_ms, "errors": [float(e) for e in errors],
            "analyze_ms": stats["analyze_ms"], "parse_ms": stats["parse_ms"],
            "launches_per_step": stats["n_launches_per_iter"], "supernodes": stats["n_supernodes"],
            "factor_flops": 2 * stats["factor_flops"], "algorithmic_bytes_per_step": sum(class_bytes.values()),
            "big_fronts": stats["n_big_fronts"], "max_front": stats["max_front"],
            "roofline": roofline, "mfma_kernel": mfma_kernel,
        }
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(args.workload)
            ref = np.array(cb.pop("errors"))
            out["cpu_baseline"] = cb
            out["chi2_final"] = float(errors[-1])
            if cb["unit"] == "GN iterations/s":   # same graph on both sides
                out["chi2_rel_diff_vs_cpu"] = abs(min(errors) - ref[-1]) / ref[-1]
                out["speedup_vs_cpu_baseline"] = value / cb["value"]
            else:
                out["speedup_vs_cpu_baseline_edges_iters"] = out["edges_iters_per_s"] / cb["edges_iters_per_s"]
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
