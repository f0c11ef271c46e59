"""Diagnostic: least-squares fit of the per-front cost model of symbolic.cpp (front_cost_us) to the in-kernel stamps."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustrobotics_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'rustrobotics_amd', 'librr_pgo_stamps.so')
from rustrobotics_amd import PoseGraph
rows = []
for name in sys.argv[1:]:
    g = PoseGraph.new(os.path.join(ROOT, 'tests/golden/g2o', name + '.g2o'))
    g.iterate_async(3); g.sync()
    L = _lib.load()
    n = C.c_int32()
    L.rr_pgo_debug_stamps(g._h, None, C.byref(n))
    out = np.zeros((n.value, 16))
    L.rr_pgo_debug_stamps(g._h, out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(n))
    tot = (out[:, 11] - out[:, 5]) * 0.01
    ok = tot > 0
    rows.append(np.column_stack([out[ok, 2], out[ok, 3], out[ok, 4], tot[ok]]))
d = np.vstack(rows)
nc, nr, kids, t = d.T
A = np.column_stack([np.ones_like(nc), nc, np.ceil(nc / 16), kids, nc * (nr + 1) ** 2, (nc + nr + 1) * nc + (nr + 1) ** 2 / 2])
coef, *_ = np.linalg.lstsq(A, t, rcond=None)
print('fronts', len(t), 'coef [1, nc, blocks, kids, nc*nu^2, elems]:', coef)
print('rms resid', np.sqrt(np.mean((A @ coef - t) ** 2)), 'mean', t.mean())
old = 3.0 + 0.45 * nc + 2e-5 * nc * (nr + 1) ** 2
print('old model rms resid', np.sqrt(np.mean((old - t) ** 2)), 'bias', np.mean(old - t))
