"""debug: staged one-rank handle and plain handle on a small lattice; usage: gpu_shard_debug2.py W H prec"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustrobotics_amd import PoseGraph, synthetic_grid_arrays, sharding
W, H, prec = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
arrays = synthetic_grid_arrays(W, H)
ref = PoseGraph.from_arrays(*arrays, precision=prec)
print("plain ", ref.stats()["n_big_fronts"], ref.stats()["n_launches_per_iter"], ref.optimize(4))
g = PoseGraph.from_arrays(*arrays, precision=prec, sharded=True)
class NoColl:
    def all_gather_boundary(self): pass
    def all_reduce_scalars(self): pass
print("staged", sharding.gauss_newton([g], 4, NoColl()))
shards, coll = sharding.emulate(arrays, 2, prec)
for it in range(2):
    shards[0].stage(0); shards[0].sync(); print("s0 r0 ok")
    shards[1].stage(0); shards[1].sync(); print("s0 r1 ok")
    coll.all_gather_boundary()
    shards[0].stage(1); shards[0].sync(); print("s1 r0 ok")
    shards[1].stage(1); shards[1].sync(); print("s1 r1 ok")
    coll.all_reduce_scalars()
    print(shards[0].stage_scalars())
