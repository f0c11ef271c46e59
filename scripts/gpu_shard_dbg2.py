"""Diagnostic (r05, the rank-dependent shared schedule): sphere2500 over P emulated ranks, iteration by iteration -- distance of the gathered
state from the unsharded handle's, by owner, and the spread of the ranks' copies of the shared poses (must be 0.0: every rank computes them
itself, bit for bit alike).  usage: gpu_shard_dbg2.py [P]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rustrobotics_amd import PoseGraph, sharding
p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/g2o/sphere2500.g2o")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
def qd(a, b):
    a, b = a.reshape(-1, 7), b.reshape(-1, 7)
    return np.maximum(np.abs(a[:, :3] - b[:, :3]).max(1), np.minimum(np.abs(a[:, 3:] - b[:, 3:]).max(1), np.abs(a[:, 3:] + b[:, 3:]).max(1)))
g = PoseGraph.new(p)
ref = PoseGraph.new(p)
shards, coll = sharding.emulate(ref.graph_arrays(), P)
owner = np.asarray(shards[0].node_owner())
for k in range(1, 9):
    g.optimize(1)
    e, n = sharding.gauss_newton(shards, 1, coll)
    d = qd(sharding.gather_state(shards), np.asarray(g.state()))
    i = int(np.argmax(d))
    print("after %d iterations: max diff %.2e at node %d (owner %d), |dx| %.3e, nodes > 1e-10: %d; by owner: %s" % (k, d.max(), i, owner[i], n[-1], int((d > 1e-10).sum()),
          " ".join("%d:%.1e" % (o, d[owner == o].max()) for o in np.unique(owner))))
# spread of the shared nodes' copies across ranks, per iteration, on fresh shards
shards, coll = sharding.emulate(ref.graph_arrays(), P)
sh = np.where(owner < 0)[0]
for k in range(1, 7):
    sharding.gauss_newton(shards, 1, coll)
    st = [np.asarray(s.state()).reshape(-1, 7) for s in shards]
    spread = [np.abs(st[r][sh] - st[0][sh]).max() for r in range(P)]
    print("iteration", k, "spread of shared copies vs rank 0:", " ".join("%.1e" % s for s in spread))
