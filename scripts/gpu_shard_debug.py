"""debug: one emulated sharded Gauss-Newton run; usage: gpu_shard_debug.py P W H prec"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustrobotics_amd import PoseGraph, synthetic_grid_arrays, sharding
P, W, H, prec = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
arrays = synthetic_grid_arrays(W, H)
shards, coll = sharding.emulate(arrays, P, prec)
print([ (g.stats()["n_big_fronts"], g.stats()["n_launches_per_iter"], g.stats()["n_levels"]) for g in shards])
try:
    errors, norms = sharding.gauss_newton(shards, 6, coll)
    print("errors", errors)
except Exception as e:
    print("FAILED", e)
ref = PoseGraph.from_arrays(*arrays, precision=prec)
print("ref   ", ref.optimize(6))
