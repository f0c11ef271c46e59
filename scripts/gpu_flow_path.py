"""Diagnostic (stamps build): the critical path through the front tree of the dataflow factorisation (k_factor_flow),
front by front and phase by phase.  usage: gpu_flow_path.py [intel|input_M3500_g2o|dlr]  (RR_PGO_LDS_FLOW=0: the level schedule)"""
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
L.rr_pgo_debug_stamps12(g._h, None, C.byref(n))
S = n.value
out = np.zeros((S, 20))
L.rr_pgo_debug_stamps12(g._h, out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(n))
st = out[:, :12] * 0.01          # us
parent = out[:, 12].astype(int); task = out[:, 13].astype(int)
nc = out[:, 14].astype(int); nr = out[:, 15].astype(int)
flow = bool(out[0, 17])
t0 = st[:, 0].min()
kids = {}
for s in range(S):
    if parent[s] >= 0: kids.setdefault(parent[s], []).append(s)
done = st[:, 11] if flow else st[:, 6]      # when a front's update matrix is out (flow: flag set)
end = st[:, 6]
root = int(np.argmax(end))
print(f'{name}: {S} fronts, {len(set(task))} tasks, dataflow={flow}; factorisation span {end.max() - t0:.1f} us (first front start -> last front end)')
# walk the critical path from the last front to finish: at every front the child that was ready last
path = []
s = root
while True:
    path.append(s)
    ks = kids.get(s, [])
    if not ks: break
    s = max(ks, key=lambda c: done[c])
path.reverse()
hdr = 'front   nc   nr kids task | start   (rel) | pre: zero+asm  extadd(all)  of which waiting | after last child: ->extadd end  panel  schur  store->flag | flag at'
print(hdr)
tot = dict(pre=0.0, wait=0.0, hop=0.0, panel=0.0, schur=0.0, store=0.0, serial=0.0)
prev_done = None
for s in path:
    ks = kids.get(s, [])
    last_child = max((done[c] for c in ks), default=None)
    zero_asm = st[s, 2] - st[s, 0]
    ext = st[s, 3] - st[s, 2]
    wait = st[s, 10] * 1.0 if flow else 0.0     # slot 10 holds accumulated wait ticks (already scaled)
    panel = st[s, 4] - st[s, 3]; schur = st[s, 5] - st[s, 4]
    store = (done[s] - st[s, 5])
    if last_child is not None and last_child > st[s, 0]:
        hop = st[s, 3] - last_child      # last child's flag -> this front's extend-add done
        tot['hop'] += hop
    else:
        hop = float('nan')
        tot['serial'] += st[s, 3] - st[s, 0]   # nothing overlapped: the whole pre-work is on the path
    tot['panel'] += panel; tot['schur'] += schur; tot['store'] += store; tot['wait'] += wait
    print(f'{s:5d} {nc[s]:4d} {nr[s]:4d} {len(ks):4d} {task[s]:4d} | {st[s,0]-t0:7.1f} | {zero_asm:6.1f} {ext:6.1f} {wait:6.1f} | {hop:6.1f} {panel:6.1f} {schur:6.1f} {store:6.1f} | {done[s]-t0:7.1f}')
print('critical path sums (us):', {k: round(v, 1) for k, v in tot.items()}, ' total', round(sum(v for k, v in tot.items() if k != 'wait'), 1))
print('  hop = last child\'s flag -> extend-add of that child done (poll + sc1 loads + LDS adds + barrier); serial = pre-work of fronts whose children were all done before they started')
# per-phase means over all fronts
ph = np.diff(st[:, 0:7], axis=1)
print('all fronts, mean us per phase [zero, asm, extadd, panel, schur, store]:', np.round(ph.mean(0), 2))
