"""TEST HELPER: one rank of a world_size-2 sharded pose graph in a process of its own (tests/test_gpu_parity.py starts two of
these with subprocess, both on GPU 0).  Real handles with opt.world_size = 2, the stage protocol of include/rr_pgo.h, the two
collectives through gloo on host memory (rustrobotics_amd.sharding.HostStagedCollectives).
usage: shard_worker.py RANK WORLD PORT MODE OUT.json      MODE: lattice | notspd"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, mode, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rustrobotics_amd import PoseGraph, PoseGraphError, sharding, synthetic_grid_arrays
    arrays = list(synthetic_grid_arrays(60, 40))
    if mode == "notspd":
        # a detached pair of poses next to the lattice, joined by one edge, reached by no prior: singular, inside ONE rank's subtree
        nk, ns, ek, ef, et, em, ei = arrays
        n = len(nk)
        arrays = [np.concatenate([nk, np.zeros(2, np.int32)]), np.concatenate([ns, [70.0, 50.0, 0.0, 71.0, 50.0, 0.0]]),
                  np.concatenate([ek, np.zeros(1, np.int32)]), np.concatenate([ef, [n]]).astype(np.int32),
                  np.concatenate([et, [n + 1]]).astype(np.int32), np.concatenate([em, [1.5, 0.25, 0.0]]),
                  np.concatenate([ei, [1.0, 0, 0, 1.0, 0, 1.0]])]
    g = PoseGraph.from_arrays(*arrays, precision="f64", device=0, rank=rank, world_size=world, sharded=True)
    coll = sharding.HostStagedCollectives(torch, dist, g)
    res = {"rank": rank, "owner": [int(o) for o in g.node_owner()]}
    s0 = np.array(g.state())
    if mode == "lattice":
        errors, norms = sharding.gauss_newton([g], 10, coll)
        res.update(errors=[float(e) for e in errors], norms=[float(x) for x in norms], state=[float(x) for x in g.state()],
                   gathered_bytes=int(coll.bytes_moved))
    else:
        g.stage(0)
        coll.all_gather_boundary()
        g.stage(1)
        coll.all_reduce_scalars()
        try:
            g.stage_scalars()
            res["code"] = 0
        except PoseGraphError as e:
            res["code"] = e.code
        res["state_unchanged"] = bool(np.array_equal(np.array(g.state()), s0))
    with open(out, "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
