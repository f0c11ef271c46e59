"""A/B of environment switches (read when a handle is created): GN it/s, launches per iteration, per-class times, and whether chi2 list
+ state after ITERS iterations are bit-identical to the FIRST configuration.
usage: python scripts/gpu_env_ab.py WORKLOAD PRECISION CONFIG [CONFIG ...]     WORKLOAD: <g2o name> | grid:WxH[:E]
       CONFIG: comma-separated NAME=VALUE pairs, or "-" for the defaults"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rustrobotics_amd import PoseGraph, synthetic_grid_arrays
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
wl, prec, configs = sys.argv[1], sys.argv[2], sys.argv[3:]
arrays = None
if wl.startswith("grid:"):
    p = wl.split(":")
    w, h = (int(x) for x in p[1].split("x"))
    arrays = synthetic_grid_arrays(w, h, int(p[2]) if len(p) > 2 else 0)
big = arrays is not None and len(arrays[0]) > 50000
ref = None
for cfg in configs:
    env = dict(kv.split("=", 1) for kv in cfg.split(",")) if cfg != "-" else {}
    env = {k: v.replace("/", ",") for k, v in env.items()}   # a comma inside a value is written "/" (RR_PGO_MERGE_CHAIN=96/1)
    for k, v in env.items():
        os.environ[k] = v
    g = PoseGraph.from_arrays(*arrays, precision=prec) if arrays is not None else PoseGraph.new(os.path.join(ROOT, "tests", "golden", "g2o", wl + ".g2o"), precision=prec)
    for k in env:
        del os.environ[k]
    s0 = np.array(g.state())
    e = g.optimize(3 if big else 4)
    h = hashlib.sha256(np.asarray(g.state()).tobytes() + np.asarray(e).tobytes()).hexdigest()[:16]
    ref = ref or h
    n = 20 if big else 50
    g.set_state(s0); g.iterate_async(5); g.sync()
    best = 1e9
    for _ in range(4):
        g.set_state(s0)
        t0 = time.perf_counter(); g.iterate_async(n); g.sync()
        best = min(best, (time.perf_counter() - t0) / n)
    g.set_state(s0)
    prof = g.profile(3)
    cls = {k: round(v[0] / 3 * 1e3, 1) for k, v in prof.items() if v[1]}
    print("%-14s %-5s %-40s %8.1f it/s (%.3f ms)  launches %2d  bits %s %s  classes(us) %s" % (
        wl, prec, cfg, 1 / best, best * 1e3, g.stats()["n_launches_per_iter"], h, "== first" if h == ref else "DIFFERS", cls), flush=True)
    del g
