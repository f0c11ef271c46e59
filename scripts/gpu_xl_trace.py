"""Diagnostic: the cross-level k_big_flow launch front by front (-DRRPGO_FLOW_TRACE build, make ../librr_pgo_trace.so): when a
front's BUILD tasks were drawn / had their children / were done, its first diagonal block, its last PANEL step, its last UPDATE
task -- and the critical path from the root back through the child that finished last.  usage: gpu_xl_trace.py [name] [precision]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustrobotics_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'rustrobotics_amd', 'librr_pgo_trace.so')
from rustrobotics_amd import PoseGraph
name = sys.argv[1] if len(sys.argv) > 1 else 'sphere2500'
prec = sys.argv[2] if len(sys.argv) > 2 else 'f64'
g = PoseGraph.new(os.path.join(ROOT, 'tests/golden/g2o', name + '.g2o'), precision=prec)
g.iterate_async(3); g.sync()
L = _lib.load()
L.rr_pgo_debug_flow_trace.restype = C.c_int64
L.rr_pgo_debug_flow_trace.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
nf, est = C.c_int32(), C.c_double()
n = L.rr_pgo_debug_flow_trace(g._h, 0, None, None, 0, C.byref(nf), C.byref(est))
tasks = np.zeros((n, 4), np.int32); st = np.zeros((n, 4, 4), np.uint64)
L.rr_pgo_debug_flow_trace(g._h, 0, tasks.ctypes.data, st.ctypes.data, n, C.byref(nf), C.byref(est))
L.rr_pgo_debug_sn_info.argtypes = [C.c_void_p, C.c_void_p]
S = L.rr_pgo_debug_sn_info(g._h, None)
info = np.zeros((S, 4), np.int32); L.rr_pgo_debug_sn_info(g._h, info.ctypes.data)
t = st.astype(np.float64) * 0.01
t0 = t[:, :, 0][st[:, :, 0] > 0].min()
t = np.where(st > 0, t - t0, np.nan)
kind, sn = tasks[:, 0] >> 24, tasks[:, 0] & 0xffffff
print(f'{name} {prec}: {n} tasks, {len(set(sn))} fronts, launch span {np.nanmax(t):.1f} us')
fin = {}
rows = {}
for f in sorted(set(sn)):
    m = sn == f
    b, p, u, d = m & (kind == 3), m & (kind == 0), m & (kind == 1), m & (kind == 2)
    r = dict(nc=info[f, 1], nr=info[f, 2], step=info[f, 3], parent=info[f, 0],
             b_taken=np.nanmin(t[b][:, :, 0]), b_kids=np.nanmax(t[b][:, :, 1]), b_done=np.nanmax(t[b][:, :, 3]),
             d_done=np.nanmax(t[d][:, 0, 3]) if d.any() else np.nan,
             p_done=np.nanmax(t[p][:, :, 3]) if p.any() else np.nan,
             u_done=np.nanmax(t[u][:, :, 3]) if u.any() else np.nan, n_p=int(p.sum()), n_u=int(u.sum()), n_b=int(b.sum()))
    r['fin'] = np.nanmax([r['p_done'], r['u_done'], r['d_done']])
    rows[f] = r
print('front  step   nc   nr  tasks(b/p/u) | build: drawn  kids-done  built | diag0   last-panel  last-update | parent')
for f, r in rows.items():
    print(f"{f:5d} {r['step']:4d} {r['nc']:5d} {r['nr']:5d}  {r['n_b']:3d}/{r['n_p']:3d}/{r['n_u']:4d} | {r['b_taken']:8.1f} {r['b_kids']:8.1f} {r['b_done']:8.1f} | {r['d_done']:8.1f} {r['p_done']:8.1f} {r['u_done']:8.1f} | {r['parent']}")
# critical path: from the front that finished last back through the child whose last update came latest
last = max(rows, key=lambda f: rows[f]['fin'])
path = [last]
while True:
    kids = [c for c in rows if rows[c]['parent'] == path[-1]]
    if not kids: break
    path.append(max(kids, key=lambda c: rows[c]['fin']))
print('critical path (root first): front, nc, nr | children done -> built -> diag0 -> panels done -> updates done  (us, and the gaps)')
for f in path:
    r = rows[f]
    print(f"  {f:5d} nc={r['nc']:4d} nr={r['nr']:5d} | kids {r['b_kids']:7.1f}  built {r['b_done']:7.1f} (+{r['b_done'] - r['b_kids']:.1f})  diag0 {r['d_done']:7.1f} (+{r['d_done'] - r['b_done']:.1f})  panels {r['p_done']:7.1f} (+{r['p_done'] - r['d_done']:.1f}, {(r['nc'] + 31) // 32} steps)  updates {r['u_done']:7.1f} (+{r['u_done'] - r['p_done']:.1f})")
