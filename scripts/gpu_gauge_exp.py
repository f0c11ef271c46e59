"""Convergence of Gauss-Newton on a lattice by precision, with and without the gauge transfer.
usage: python scripts/gpu_gauge_exp.py WxH[:E] [iters]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rustrobotics_amd import PoseGraph

spec = sys.argv[1] if len(sys.argv) > 1 else "400x250:1000000"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 15
parts = spec.split(":")
w, h = (int(x) for x in parts[0].split("x"))
e = int(parts[1]) if len(parts) > 1 else 0
out = {}
ref_state = None
for prec, gauge in [("f64", "1"), ("mixed", "1"), ("mixed", "0"), ("f32", "1"), ("f32", "0")]:
    os.environ["RR_PGO_GAUGE"] = gauge
    g = PoseGraph.synthetic_grid(w, h, e, precision=prec)
    t0 = time.time()
    errs, norms = g.optimize(iters, return_norms=True)
    dt = time.time() - t0
    st = np.array(g.state())
    if ref_state is None:
        ref_state = st
    d = st - ref_state
    d[2::3] = (d[2::3] + np.pi) % (2 * np.pi) - np.pi
    key = f"{prec}/gauge{gauge}"
    out[key] = {"errors": [float(x) for x in errs], "norms": [float(x) for x in norms], "seconds": dt,
                "max_pose_diff_vs_f64": float(np.abs(d).max())}
    print(key, "iters", len(norms), "chi2", ["%.9g" % x for x in errs[-3:]], "norms", ["%.3g" % x for x in norms], "posediff %.3g" % np.abs(d).max(), flush=True)
    del g
json.dump(out, open(os.path.join("gpurun_out", "gauge_exp_%s.json" % spec.replace(":", "_")), "w"), indent=1)
