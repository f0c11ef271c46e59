"""A/B of the Schur pass beside the flow launch (k_big_schur_flow, RR_PGO_SCHUR_OVERLAP=<nf>): ms per Gauss-Newton step of the
1M-edge lattice for every limit, and whether the state after three iterations is bit-identical to the limit 0 (k_big_schur
behind the flow launch).  usage: python scripts/gpu_overlap_ab.py [precision] [limits...] [--grid G]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rustrobotics_amd import PoseGraph, synthetic_grid_arrays

args = [a for a in sys.argv[1:] if not a.startswith("--")]
prec = args[0] if args else "f32"
limits = [int(x) for x in args[1:]] or [0, 2, 4, 8, 16, 32, 64, 128]
arrays = synthetic_grid_arrays(400, 250, 1000000)
ref = None
for lim in limits:
    os.environ["RR_PGO_SCHUR_OVERLAP"] = str(lim)
    g = PoseGraph.from_arrays(*arrays, precision=prec)
    s0 = np.array(g.state())
    g.iterate_async(3); g.sync()
    h = hashlib.sha256(np.asarray(g.state()).tobytes()).hexdigest()[:16]
    if ref is None:
        ref = h
    g.set_state(s0)
    g.iterate_async(5); g.sync()
    best = 1e9
    for _ in range(3):
        g.set_state(s0)
        t0 = time.perf_counter(); g.iterate_async(20); g.sync()
        best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
    g.set_state(s0)
    prof = g.profile(3)
    cls = {k: round(v[0] / 3 * 1e3) for k, v in prof.items() if v[1]}
    print("overlap nf<=%-4d %.3f ms/step  state %s %s  launches %d  classes(us) %s" % (
        lim, best, h, "== ref" if h == ref else "DIFFERS", g.stats()["n_launches_per_iter"], cls), flush=True)
    del g
