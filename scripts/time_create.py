import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rustrobotics_amd import PoseGraph, synthetic_grid_arrays
arrays = synthetic_grid_arrays(400, 250, 1000000)
for rep in range(2):
    t0 = time.perf_counter(); g = PoseGraph.from_arrays(*arrays, precision='f32'); t1 = time.perf_counter()
    print(os.environ.get('RR_PGO_FLOW', 'default'), 'create %.0f ms' % ((t1 - t0) * 1e3), 'analyze', round(g.stats()['analyze_ms']))
    del g
