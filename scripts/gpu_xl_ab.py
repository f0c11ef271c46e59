"""A/B of the cross-level form of k_big_flow (RR_PGO_FLOW_XL=0: one build + one flow launch per level): it/s of the small graphs with
fronts beyond LDS, bit-identity of the state after four iterations, launches per iteration.  usage: python scripts/gpu_xl_ab.py [names...]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rustrobotics_amd import PoseGraph
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = sys.argv[1:] or ["sphere2500", "torus3D"]
for name in names:
    path = os.path.join(ROOT, "tests", "golden", "g2o", name + ".g2o")
    ref = None
    for xl in ("0", None):
        if xl is None:
            os.environ.pop("RR_PGO_FLOW_XL", None)
        else:
            os.environ["RR_PGO_FLOW_XL"] = xl
        g = PoseGraph.new(path)
        s0 = np.array(g.state())
        e = g.optimize(4)
        h = hashlib.sha256(np.asarray(g.state()).tobytes() + np.asarray(e).tobytes()).hexdigest()[:16]
        ref = ref or h
        g.set_state(s0); g.iterate_async(20); g.sync()
        best = 1e9
        for _ in range(5):
            g.set_state(s0)
            t0 = time.perf_counter(); g.iterate_async(50); g.sync()
            best = min(best, (time.perf_counter() - t0) / 50)
        g.set_state(s0)
        prof = g.profile(5)
        cls = {k: round(v[0] / 5 * 1e3, 1) for k, v in prof.items() if v[1]}
        print("%-12s XL=%-4s %7.1f it/s (%.3f ms)  launches %2d  chi2 %s  bits %s %s  classes(us) %s" % (
            name, "on" if xl is None else "off", 1 / best, best * 1e3, g.stats()["n_launches_per_iter"], e[-1], h, "== ref" if h == ref else "DIFFERS", cls), flush=True)
        del g
