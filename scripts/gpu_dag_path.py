"""Diagnostic (stamps build): per-task durations and the longest dependency chain of the LDS-front task DAG, against the
sum of the per-level worst tasks (what the level-synchronous launches pay)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
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
S = n.value
step, task, parent = out[:, 0].astype(int), out[:, 1].astype(int), out[:, 15].astype(int)
dur = {}
for t in set(task):
    if t < 0: continue
    m = task == t
    dur[t] = (out[m, 11].max() - out[m, 5].min()) * 0.01
tstep = {t: step[task == t][0] for t in dur}
# task DAG: task of a front -> task of its parent front
ptask = {}
for s in range(S):
    if task[s] < 0 or parent[s] < 0 or task[parent[s]] < 0: continue
    if task[parent[s]] != task[s]: ptask.setdefault(task[s], set()).add(task[parent[s]])
order = sorted(dur, key=lambda t: (tstep[t], t))
finish = {}
kids = {}
for t, ps in ptask.items():
    for p in ps: kids.setdefault(p, []).append(t)
for t in order:
    finish[t] = dur[t] + max([finish[c] for c in kids.get(t, [])], default=0.0)
levels = {}
for t in dur: levels[tstep[t]] = max(levels.get(tstep[t], 0), dur[t])
print(name, 'tasks', len(dur), 'levels', len(levels), 'sum of per-level worst %.1f us' % sum(levels.values()), 'longest chain %.1f us' % max(finish.values()))
print('per-level worst', {k: round(v, 1) for k, v in sorted(levels.items())})
t = max(finish, key=finish.get); chain = []
while True:
    chain.append((t, tstep[t], round(dur[t], 1)))
    ks = kids.get(t, [])
    if not ks: break
    t = max(ks, key=lambda c: finish[c])
print('chain (task, level, us):', chain)
# front-granularity dataflow: every front its own workgroup, h us of synchronisation per tree edge
fdur = (out[:, 11] - out[:, 5]) * 0.01
valid = task >= 0
ch = {}
for s in range(S):
    if valid[s] and parent[s] >= 0 and valid[parent[s]]: ch.setdefault(parent[s], []).append(s)
for h in (0.0, 1.0, 2.0, 3.0):
    fin = np.zeros(S)
    for s in range(S):   # children precede parents
        if valid[s]: fin[s] = fdur[s] + max([fin[c] + h for c in ch.get(s, [])], default=0.0)
    s = int(np.argmax(fin)); depth = 0; t = s
    while ch.get(t): t = max(ch[t], key=lambda c: fin[c]); depth += 1
    print('front-level chain with %.0f us per edge: %.1f us over %d fronts' % (h, fin.max(), depth + 1))
