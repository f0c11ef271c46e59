"""Synthetic lattice checks on the GPU: f64/f32 GPU vs oracle (small), timings (large)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustrobotics_amd import PoseGraph, synthetic_grid_arrays
from oracle.oracle import OracleGraph
W, H = int(sys.argv[1]), int(sys.argv[2])
E = int(sys.argv[3]) if len(sys.argv) > 3 else 0
prec = sys.argv[4] if len(sys.argv) > 4 else 'f64'
t = time.time(); arrays = synthetic_grid_arrays(W, H, E); print('generate %.2f s' % (time.time() - t))
t = time.time(); g = PoseGraph.from_arrays(*arrays, precision=prec); print('create %.2f s' % (time.time() - t))
st = g.stats(); print({k: st[k] for k in ('n_supernodes', 'n_levels', 'n_launches_per_iter', 'max_front', 'n_big_fronts', 'factor_flops', 'analyze_ms', 'nnz_l_scalars')})
t = time.time(); e = g.optimize(10); dt = time.time() - t
print(prec, 'errors', ['%.8g' % x for x in e], 'optimize wall %.1f ms' % (dt * 1e3))
if len(arrays[0]) <= 12000 and os.environ.get('ORACLE', '1') == '1':
    o = OracleGraph.from_arrays(*arrays); t = time.time(); eo = o.optimize(10); print('oracle errors', ['%.8g' % x for x in eo], 'wall %.1f ms' % ((time.time() - t) * 1e3))
    print('state max abs diff', np.abs(g.state() - o.state()).max())
g2 = PoseGraph.from_arrays(*arrays, precision=prec)
g2.iterate_async(2); g2.sync()
t = time.time(); g2.iterate_async(5); g2.sync(); dt = time.time() - t
print('iterate: %.2f ms / GN iteration' % (dt / 5 * 1e3))
print('profile(3):', {k: (round(v[0] / 3, 3), v[1] // 3) for k, v in g2.profile(3).items()})
