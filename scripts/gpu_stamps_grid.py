"""Diagnostic: distribution of the LDS-front task durations of a lattice (-DRRPGO_STAMPS build): per step of the task
schedule the tasks' durations, their start times (dispatch rounds) and how full the CUs were.  usage: gpu_stamps_grid.py W H E [precision]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustrobotics_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'rustrobotics_amd', 'librr_pgo_stamps.so')
from rustrobotics_amd import PoseGraph, synthetic_grid_arrays
w, h, e = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
prec = sys.argv[4] if len(sys.argv) > 4 else 'f32'
g = PoseGraph.from_arrays(*synthetic_grid_arrays(w, h, e), precision=prec)
g.iterate_async(3); g.sync()
L = _lib.load()
n = C.c_int32()
L.rr_pgo_debug_stamps(g._h, None, C.byref(n))
out = np.zeros((n.value, 16))
L.rr_pgo_debug_stamps(g._h, out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(n))
steps = out[:, 0].astype(int)
for st in sorted(set(steps)):
    m = (steps == st) & (out[:, 5] > 0)
    if not m.any():
        continue
    tasks = out[m, 1].astype(int)
    t0 = out[m, 5].min()
    start, end = {}, {}
    for t, a, b in zip(tasks, out[m, 5], out[m, 11]):
        start[t] = min(start.get(t, 1e30), a); end[t] = max(end.get(t, 0), b)
    ids = np.array(sorted(start))
    s = np.array([start[t] - t0 for t in ids]) * 0.01
    d = np.array([end[t] - start[t] for t in ids]) * 0.01
    span = (max(end.values()) - t0) * 0.01
    print(f'step {st}: {len(ids)} tasks, {m.sum()} fronts; span {span:.1f} us; task duration mean {d.mean():.1f} p10 {np.percentile(d,10):.1f} p50 {np.median(d):.1f} '
          f'p90 {np.percentile(d,90):.1f} max {d.max():.1f}; sum of durations / (256 CUs x span) = {d.sum() / (256 * span):.2f}')
    # how many tasks are running over time
    ts = np.linspace(0, span, 21)[:-1]
    running = [(int(((s <= x) & (s + d > x)).sum())) for x in ts]
    print('   tasks running at 5 % steps of the span:', running)
    order = np.argsort(ids)
    print('   duration by task index (deciles of the index range):', [round(float(d[order][i:i + max(len(ids) // 10, 1)].mean()), 1) for i in range(0, len(ids), max(len(ids) // 10, 1))][:10])
    worst = ids[np.argmax(d)]
    mm = m & (out[:, 1] == worst)
    print(f'   longest task {worst}: fronts (nc, nr, kids, us):', [(int(r[2]), int(r[3]), int(r[4]), round((r[11] - r[5]) * 0.01, 1)) for r in out[mm]])
    top = np.argsort(-d)[:8]
    print('   eight longest tasks (us, fronts):', [(round(float(d[i]), 1), int((m & (out[:, 1] == ids[i])).sum())) for i in top])
    names = ['zero', 'asm', 'extadd', 'panel', 'schur', 'store']
    for r in out[mm]:
        print('      nc', int(r[2]), 'nr', int(r[3]), {k: round(float(v) * 0.01, 1) for k, v in zip(names, np.diff(r[5:12]))}, 'p.diag/trsm/upd', [round(float(v) * 0.01, 1) for v in r[12:15]])
    allph = np.diff(out[m][:, 5:12], axis=1).sum(0) * 0.01
    print('   all fronts of the step, phase sums (us):', {k: round(float(v)) for k, v in zip(names, allph)}, 'p.diag/trsm/upd', [round(float(v) * 0.01) for v in out[m][:, 12:15].sum(0)])
