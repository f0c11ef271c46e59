"""Emulate P ranks of the sharded path on ONE GPU (sequentially) and time each rank's stages:
projected multi-GPU step = max_r(stage0) + max_r(stage1) + stage2 + 2 all-reduces."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustrobotics_amd import PoseGraph, synthetic_grid_arrays
W, H, E, prec = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
arrays = synthetic_grid_arrays(W, H, E)
dt = torch.float64 if prec == 'f64' else torch.float32
for P in [int(x) for x in sys.argv[5:]]:
    shards = [PoseGraph.from_arrays(*arrays, precision=prec, rank=r, world_size=P) for r in range(P)]
    bufs = {0: [], 1: []}
    for g in shards:
        for which in (0, 1):
            _, n, es = g.exchange_info(which)
            t = torch.zeros(max(n, 1), dtype=dt, device='cuda'); g.bind_exchange(which, t.data_ptr(), t.numel()); bufs[which].append(t)
    def allreduce(which):
        tot = bufs[which][0].clone()
        for t in bufs[which][1:]: tot += t
        for t in bufs[which]: t.copy_(tot)
        torch.cuda.synchronize()
    times = np.zeros((3, P)); errs = []
    for it in range(3):
        for stage in (0, 1, 2):
            for r, g in enumerate(shards):
                g.sync(); torch.cuda.synchronize(); t0 = time.perf_counter(); g.stage(stage); g.sync(); dtm = time.perf_counter() - t0
                if it > 0: times[stage, r] += dtm / 2
            if stage < 2: allreduce(stage)
        errs.append(shards[0].stage_scalars()[0])
    xb = [bufs[w][0].numel() * bufs[w][0].element_size() for w in (0, 1)]
    proj = times[0].max() + times[1].max() + times[2].max()
    print(f'P={P}: stage0 max {times[0].max()*1e3:.2f} ms (mean {times[0].mean()*1e3:.2f}), stage1 {times[1].max()*1e3:.2f}, stage2 {times[2].max()*1e3:.2f} '
          f'-> compute {proj*1e3:.2f} ms/iter ; exchange {xb[0]/1e6:.1f} MB + {xb[1]/1e6:.1f} MB ; chi2 {errs}')
    del shards, bufs
