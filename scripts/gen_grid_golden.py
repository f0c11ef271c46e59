#!/usr/bin/env python3
"""Generates tests/golden/grid400x250.json: the CPU oracle's Gauss-Newton run on BASELINE config 4
(the 400 x 250 lattice, 100 000 poses / 1 000 000 edges).

The reference cannot run this configuration itself (its COO has capacity len^2,
pose_graph_optimization.rs:113-119), so the fp64 oracle -- pinned to the reference's own goldens on the
small files -- IS the reference here.  One linearize+solve takes a few minutes on one core and ~1.5 GB;
the whole run about half an hour.  Run it once, in the build container:

    python scripts/gen_grid_golden.py [W H E]          # default 400 250 1000000

The fixture holds: chi2 per iteration up to the |dx| < 1e-4 stop (:298-300), |dx| per iteration, the first
step dx and the final state at SAMPLE nodes, the anchor, and a digest of the generated graph so that a test
can tell a generator drift from a solver drift.
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.oracle import OracleGraph  # noqa: E402
from rustrobotics_amd import synthetic_grid_arrays  # noqa: E402  (host-only generator, no GPU needed)


def sample_nodes(n, count=64):
    """anchor (node 0), the last node, and evenly spaced ones in between"""
    idx = sorted(set([0, 1, n - 1] + [int(round(t)) for t in np.linspace(0, n - 1, count)]))
    return idx


def graph_digest(arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def main():
    w, hgt, e = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (400, 250, 1000000)
    out_path = os.path.join(ROOT, "tests", "golden", f"grid{w}x{hgt}.json")
    arrays = synthetic_grid_arrays(w, hgt, e)
    g = OracleGraph.from_arrays(*arrays)
    n = g.num_nodes
    nodes = sample_nodes(n)
    t0 = time.time()
    chi0 = g.global_error()
    print(f"[{time.time() - t0:7.1f}s] chi2_0 = {chi0!r}", flush=True)
    dx0 = g.linearize_and_solve()          # first Gauss-Newton step from the initial state (state untouched)
    print(f"[{time.time() - t0:7.1f}s] first step |dx| = {np.linalg.norm(dx0)!r}", flush=True)
    errors, norms = [chi0], []
    # optimize(), :247-303, one iteration per call so that progress is visible; the stop rule is applied here
    for it in range(30):
        dx = dx0 if it == 0 else g.linearize_and_solve()
        g.update_nodes(dx, 1.0)
        nrm = float(np.linalg.norm(dx))
        err = g.global_error()
        norms.append(nrm)
        errors.append(err)
        print(f"[{time.time() - t0:7.1f}s] it {it}: chi2 = {err!r}  |dx| = {nrm!r}", flush=True)
        if nrm < 1e-4:
            break
    state = g.state().reshape(n, 3)
    fixture = {
        "generator": f"scripts/gen_grid_golden.py {w} {hgt} {e}",
        "workload": f"grid:{w}x{hgt}:{e}", "width": w, "height": hgt, "n_nodes": n, "n_edges": g.num_edges,
        "graph_sha256": graph_digest(arrays),
        "anchor_node": 0,
        "errors": errors, "norms": norms,
        "sample_nodes": nodes,
        "first_dx_at_samples": [[float(v) for v in dx0[3 * i:3 * i + 3]] for i in nodes],
        "final_state_at_samples": [[float(v) for v in state[i]] for i in nodes],
        "final_state_sum": [float(state[:, 0].sum()), float(state[:, 1].sum())],
        "oracle_seconds": time.time() - t0,
    }
    with open(out_path, "w") as f:
        json.dump(fixture, f, indent=1)
    print("wrote", out_path)


if __name__ == "__main__":
    main()
