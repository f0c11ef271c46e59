"""The three linearisation kernels on the 1M-edge lattice (fp32) and on intel.g2o (fp64): the pull form (the product), one thread per
edge (RR_PGO_EDGE_LINEARIZE=1), one wavefront per edge with LDS staging and an LDS-reduced scatter-add (=2: the form BASELINE.json's
north_star words).  Per-launch time of the linearisation class from rr_pgo_profile (HIP events), chi2 and the first step against the
pull form.    usage: python scripts/gpu_lin_forms.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rustrobotics_amd import PoseGraph, synthetic_grid_arrays
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for what, prec in (("lattice", "f32"), ("intel", "f64")):
    arrays = synthetic_grid_arrays(400, 250, 1000000) if what == "lattice" else None
    ref = None
    for form in ("0", "1", "2"):
        if form != "0":
            os.environ["RR_PGO_EDGE_LINEARIZE"] = form
        g = PoseGraph.from_arrays(*arrays, precision=prec) if arrays is not None else PoseGraph.new(os.path.join(ROOT, "tests/golden/g2o/intel.g2o"), precision=prec)
        os.environ.pop("RR_PGO_EDGE_LINEARIZE", None)
        chi = g.global_error()
        dx = g.linearize_and_solve()
        prof = g.profile(5)
        lin_us = 1e3 * prof["linearize"][0] / 5
        n = prof["linearize"][1] / 5
        if ref is None:
            ref = (chi, dx)
        print(f"{what:8s} {prec} form {form} ({'pull, 8 lanes per node' if form == '0' else 'thread per edge' if form == '1' else 'WAVE per edge, LDS-staged, LDS-reduced'}): "
              f"linearisation {lin_us:8.1f} us per iteration in {n:.0f} launch(es); chi2 rel diff {abs(chi - ref[0]) / ref[0]:.1e}, "
              f"first step max diff {np.abs(dx - ref[1]).max():.1e}", flush=True)
        del g
