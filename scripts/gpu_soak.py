"""Bit-identity soak: every repeat of the same Gauss-Newton run must give the same chi2 list and the same state, to the
last bit.  The design is deterministic (fixed-order sums, no atomics on data), so any difference between two repeats is a
race between workgroups (a hand-off read early), or a read of memory nothing wrote (whatever the allocation held before).

  python scripts/gpu_soak.py CONFIG[,CONFIG...] [--repeats R] [--iters K] [--rebuild N] [--poison]

CONFIG: lattice8:mixed  lattice8:f32  (400 x 250 / 1M edges over 8 emulated ranks)   lattice:mixed lattice:f32 lattice:f64
        sphere2500 torus3D parking-garage intel input_M3500_g2o dlr  (unsharded, fp64; NAME:mixed for mixed)
        sphere8 (sphere2500 over 8 emulated ranks)    grid60x40 / grid100x100:P (small lattices, P emulated ranks)
--rebuild N: drop and rebuild the handles every N repeats (a fresh allocation each time)
--poison:    before every (re)build fill most of the free device memory with junk and release it, so that the
             library's allocations do not start out zeroed
On the first difference: the iteration whose chi2 differs first, how many state entries differ, the first nodes."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rustrobotics_amd import PoseGraph, synthetic_grid_arrays, sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def poison(pattern):
    """Fill (most of) the free device memory with a pattern, then hand it back to the HIP allocator."""
    free, _ = torch.cuda.mem_get_info()
    n = int(min(free * 0.5, 48e9)) // 8
    t = torch.empty(n, dtype=torch.int64, device="cuda")
    if pattern == "rand":
        t.random_()
    else:
        t.fill_(pattern)
    torch.cuda.synchronize()
    del t
    torch.cuda.empty_cache()


def build(cfg):
    name, _, arg = cfg.partition(":")
    if name in ("lattice8", "lattice"):
        arrays = synthetic_grid_arrays(400, 250, 1000000)
        prec = arg or "mixed"
        if name == "lattice8":
            return ("sharded",) + sharding.emulate(arrays, 8, prec)
        return ("single", PoseGraph.from_arrays(*arrays, precision=prec))
    if name.startswith("grid"):
        w, h = (int(x) for x in name[4:].split("x"))
        arrays = synthetic_grid_arrays(w, h)
        P = int(arg or 1)
        if P > 1:
            return ("sharded",) + sharding.emulate(arrays, P, "f64")
        return ("single", PoseGraph.from_arrays(*arrays))
    if name == "sphere8":
        ref = PoseGraph.new(os.path.join(ROOT, "tests", "golden", "g2o", "sphere2500.g2o"))
        return ("sharded",) + sharding.emulate(ref.graph_arrays(), 8)
    return ("single", PoseGraph.new(os.path.join(ROOT, "tests", "golden", "g2o", name + ".g2o"), precision=arg or "f64"))


def run_once(h, s0, iters):
    if h[0] == "sharded":
        shards, coll = h[1], h[2]
        for g, s in zip(shards, s0):
            g.set_state(s)
        errs, norms = sharding.gauss_newton(shards, iters, coll, tolerance=0.0)
        st = sharding.gather_state(shards)
    else:
        g = h[1]
        g.set_state(s0)
        errs, norms = g.optimize(iters, return_norms=True)
        st = np.asarray(g.state())
    return np.array(errs + norms), st


def initial(h):
    if h[0] == "sharded":
        return [np.array(g.state()) for g in h[1]]
    return np.array(h[1].state())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs")
    ap.add_argument("--repeats", type=int, default=40)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--rebuild", type=int, default=0)
    ap.add_argument("--poison", action="store_true")
    a = ap.parse_args()
    bad = 0
    for cfg in a.configs.split(","):
        t0 = time.time()
        patterns = ["rand", -1, 0x7FF0000000000000, 0x3FF0000000000000]
        if a.poison:
            poison(patterns[0])
        h = build(cfg)
        s0 = initial(h)
        ref_e, ref_s = run_once(h, s0, a.iters)
        ndiff, nb = 0, 1
        for r in range(1, a.repeats):
            if a.rebuild and r % a.rebuild == 0:
                del h
                if a.poison:
                    poison(patterns[nb % len(patterns)])
                h = build(cfg)
                nb += 1
            e, s = run_once(h, s0, a.iters)
            same_e = e.tobytes() == ref_e.tobytes()
            same_s = s.tobytes() == ref_s.tobytes()
            if not (same_e and same_s):
                ndiff += 1
                if ndiff <= 5:
                    ie = np.nonzero(e.view(np.int64) != ref_e.view(np.int64))[0]
                    js = np.nonzero(s.view(np.int64) != ref_s.view(np.int64))[0]
                    print("  %s repeat %d (build %d) DIFFERS: chi2/norm entries %s (of %d: chi2 0..%d then |dx|); first: %s vs %s; "
                          "state entries differing %d of %d, first %s, max abs diff %.3e" % (
                              cfg, r, nb, ie[:8].tolist(), len(e), a.iters, e[ie[:2]].tolist(), ref_e[ie[:2]].tolist(),
                              len(js), len(s), js[:8].tolist(), float(np.nanmax(np.abs(s - ref_s))) if len(js) else 0.0), flush=True)
        print("%-18s repeats %d iters %d builds %d poison %d: %s  (chi2 %.9g -> %.9g, %.1f s)" % (
            cfg, a.repeats, a.iters, nb, int(a.poison), "ALL BIT-IDENTICAL" if ndiff == 0 else "%d DIFFERENT" % ndiff,
            ref_e[0], ref_e[a.iters], time.time() - t0), flush=True)
        bad += ndiff
        del h
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
