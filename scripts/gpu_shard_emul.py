"""Projection of the sharded lattice step from per-rank stage times measured on ONE GPU.

Each of the P emulated ranks runs its stage 0 and stage 1 alone on the GPU (HIP graph replays, host timer around a
synchronised stage); the collectives are costed, not measured: the all-gather by bytes / per-link bandwidth
(every rank sends its chunk to P-1 peers over separate xGMI links, MI355X_MICROARCH.md: 153.6 GB/s per link per
direction peak, ~50 GB/s achieved per link is the planning number used here) plus a launch latency, the two-double
all-reduce by latency only (150 us each: measured through torch.distributed on a one-rank group).  usage: python scripts/gpu_shard_emul.py [WxH[:E]] [precision] [P ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rustrobotics_amd import PoseGraph, synthetic_grid_arrays, sharding

spec = sys.argv[1] if len(sys.argv) > 1 else "400x250:1000000"
prec = sys.argv[2] if len(sys.argv) > 2 else "mixed"
Ps = [int(x) for x in sys.argv[3:]] or [2, 4, 8]
parts = spec.split(":")
w, h = (int(x) for x in parts[0].split("x"))
arrays = synthetic_grid_arrays(w, h, int(parts[1]) if len(parts) > 1 else 0)
LINK_GBPS, COLL_LAT_US = 50.0, 150.0   # 150 us per RCCL collective: what the one-rank nccl leg of bench.py costs over the emulated P = 1 run (r02: 6.41 against 6.11 ms per step, two collectives)

g = PoseGraph.from_arrays(*arrays, precision=prec)
g.iterate_async(3); g.sync()
t0 = time.perf_counter(); g.iterate_async(10); g.sync()
base_ms = (time.perf_counter() - t0) * 100
del g
out = {"workload": spec, "precision": prec, "unsharded_ms": base_ms, "link_GBps_assumed": LINK_GBPS,
       "collective_latency_us_assumed": COLL_LAT_US, "P": {}}
print("unsharded %.3f ms/step" % base_ms, flush=True)
gold = None
if spec == "400x250:1000000":
    gold = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "grid400x250.json")))["errors"]
for P in Ps:
    shards, coll = sharding.emulate(arrays, P, prec)
    # numbers of a wrong factorisation are worthless: the sharded run must first reproduce the oracle's chi2
    s_init = [np.array(gq.state()) for gq in shards]
    errs, _ = sharding.gauss_newton(shards, 8, coll, tolerance=0.0)
    if gold is not None:
        rel = abs(min(errs) - gold[-1]) / gold[-1]
        assert rel <= (1e-8 if prec != "f32" else 1e-6), (P, errs, gold[-1])
        print("P=%d chi2 %.6f vs oracle fixture %.6f (rel %.1e)" % (P, min(errs), gold[-1], rel), flush=True)
    for gq, s0 in zip(shards, s_init):
        gq.set_state(s0)
    for _ in range(2):   # warm-up: captures the stage graphs
        for gq in shards: gq.stage(0)
        coll.all_gather_boundary()
        for gq in shards: gq.stage(1)
        coll.all_reduce_scalars()
    st = np.zeros((2, P))
    reps = 5
    for _ in range(reps):
        for r, gq in enumerate(shards):
            gq.sync(); t0 = time.perf_counter(); gq.stage(0); gq.sync(); st[0, r] += time.perf_counter() - t0
        coll.all_gather_boundary()
        for r, gq in enumerate(shards):
            gq.sync(); t0 = time.perf_counter(); gq.stage(1); gq.sync(); st[1, r] += time.perf_counter() - t0
        coll.all_reduce_scalars()
    st *= 1e3 / reps
    chunk_bytes = coll.chunk * coll.xch[0].element_size()
    gather_ms = chunk_bytes / (LINK_GBPS * 1e9) * 1e3 + COLL_LAT_US * 1e-3   # P-1 links in parallel, one chunk per link
    reduce_ms = COLL_LAT_US * 1e-3
    total = st[0].max() + gather_ms + st[1].max() + reduce_ms
    out["P"][P] = {"stage0_ms_by_rank": st[0].tolist(), "stage1_ms_by_rank": st[1].tolist(), "chunk_MB": chunk_bytes / 1e6,
                   "all_gather_ms_est": gather_ms, "all_reduce_ms_est": reduce_ms, "step_ms_projected": total,
                   "speedup_projected": base_ms / total}
    print("P=%d stage0 max %.3f (min %.3f)  stage1 max %.3f  chunk %.2f MB gather %.3f  -> %.3f ms  %.2fx" %
          (P, st[0].max(), st[0].min(), st[1].max(), chunk_bytes / 1e6, gather_ms, total, base_ms / total), flush=True)
    del shards, coll
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/shard_emul_%s_%s.json" % (spec.replace(":", "_"), prec), "w"), indent=1)
