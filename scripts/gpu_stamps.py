"""Diagnostic: per-front phase times from the -DRRPGO_STAMPS build (100 MHz wall clock)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustrobotics_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'rustrobotics_amd', 'librr_pgo_stamps.so')
from rustrobotics_amd import PoseGraph
name = sys.argv[1] if len(sys.argv) > 1 else 'intel'
g = PoseGraph.new(os.path.join(ROOT, 'tests/golden/g2o', name + '.g2o'))
g.iterate_async(3); g.sync()
L = _lib.load()
n = C.c_int32()
L.rr_pgo_debug_stamps(g._h, None, C.byref(n))
out = np.zeros((n.value, 16))
L.rr_pgo_debug_stamps(g._h, out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(n))
ph = np.concatenate([np.diff(out[:, 5:12], axis=1), out[:, 12:15]], axis=1) * 0.01  # us
names = ['zero', 'asm', 'extadd', 'panel', 'schur', 'store', 'p.diag', 'p.trsm', 'p.upd']
if os.environ.get('SOLVE'):
    names = ['stage', 'gemv', 'trisolve', 'store']
    ph = np.diff(out[:, 5:10], axis=1) * 0.01
    out[:, 11] = out[:, 9]
steps = out[:, 0].astype(int)
for st in sorted(set(steps)):
    m = steps == st
    tasks = out[m, 1].astype(int)
    # per task totals
    tot = {}
    for t in set(tasks):
        mm = m & (out[:, 1] == t)
        tot[t] = (out[mm, 11].max() - out[mm, 5].min()) * 0.01
    worst = max(tot, key=tot.get)
    mm = m & (out[:, 1] == worst)
    print(f'step {st}: {m.sum()} fronts in {len(tot)} tasks; worst task {worst}: {tot[worst]:.1f} us over {mm.sum()} fronts; '
          f'mean task {np.mean(list(tot.values())):.1f} us')
    print('   worst-task phase sums (us):', {k: round(float(v), 1) for k, v in zip(names, ph[mm].sum(0))})
    idx = np.where(mm)[0]
    for s in idx[-6:]:
        print(f'     sn {s}: nc={int(out[s,2])} nr={int(out[s,3])} kids={int(out[s,4])} ', {k: round(float(v), 1) for k, v in zip(names, ph[s])})
