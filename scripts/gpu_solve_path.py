"""Diagnostic (make -C rustrobotics_amd/csrc ../librr_pgo_stamps_solve.so): the critical path of the dataflow back substitution
(k_solve_flow), front by front: root -> the leaf that finishes last.  usage: gpu_solve_path.py [intel|input_M3500_g2o|dlr]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustrobotics_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'rustrobotics_amd', 'librr_pgo_stamps_solve.so')
from rustrobotics_amd import PoseGraph
name = sys.argv[1] if len(sys.argv) > 1 else 'intel'
g = PoseGraph.new(os.path.join(ROOT, 'tests/golden/g2o', name + '.g2o'))
g.iterate_async(3); g.sync()
L = _lib.load()
n = C.c_int32()
L.rr_pgo_debug_stamps12(g._h, None, C.byref(n))
S = n.value
out = np.zeros((S, 20))
L.rr_pgo_debug_stamps12(g._h, out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(n))
st = out[:, :5] * 0.01          # us: 0 begun, 1 x[rows] gathered, 2 L21^T x done (chain operands in), 3 chain done, 4 x stored
parent = out[:, 12].astype(int); nc = out[:, 14].astype(int); nr = out[:, 15].astype(int)
kids = {}
for s in range(S):
    if parent[s] >= 0: kids.setdefault(parent[s], []).append(s)
t0 = st[:, 0].min()
# latest finish in each subtree
fin = st[:, 4].copy()
for s in range(S):          # children precede parents in numbering: accumulate upwards
    if parent[s] >= 0: fin[parent[s]] = max(fin[parent[s]], fin[s])
roots = [s for s in range(S) if parent[s] < 0]
s = max(roots, key=lambda r: fin[r])
print(f'{name}: back substitution span {st[:, 4].max() - t0:.1f} us')
print('front   nc   nr | begun | wait+gather  L21^T x  chain (per 16 columns)  store | done at')
tot = np.zeros(4)
while True:
    ph = np.diff(st[s, :5])
    nb = (nc[s] + 15) // 16
    print(f'{s:5d} {nc[s]:4d} {nr[s]:4d} | {st[s,0]-t0:6.1f} | {ph[0]:6.1f} {ph[1]:6.1f} {ph[2]:6.1f} ({ph[2]/nb:4.2f}) {ph[3]:6.1f} | {st[s,4]-t0:6.1f}')
    tot += ph
    ks = kids.get(s, [])
    if not ks: break
    s = max(ks, key=lambda c: fin[c])
print('path sums (us): wait+gather %.1f, L21^T x %.1f, chain %.1f, store+flag %.1f' % tuple(tot))
