"""Diagnostic: per-task timeline of k_big_flow from the -DRRPGO_FLOW_TRACE build (make ../librr_pgo_trace.so).
usage: gpu_flow_trace.py [grid:WxH[:E] | <g2o name>] [precision] [level ...]
Per flow level: span, the chain of front 0 (look waves and tile (0, 0)) step by step, and how busy the workgroups were."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustrobotics_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'rustrobotics_amd', 'librr_pgo_trace.so')
from rustrobotics_amd import PoseGraph, synthetic_grid_arrays

wl = sys.argv[1] if len(sys.argv) > 1 else 'grid:400x250:1000000'
prec = sys.argv[2] if len(sys.argv) > 2 else 'f32'
only = [int(x) for x in sys.argv[3:]]
if wl.startswith('grid:'):
    p = wl.split(':')
    w, h = (int(x) for x in p[1].split('x'))
    g = PoseGraph.from_arrays(*synthetic_grid_arrays(w, h, int(p[2]) if len(p) > 2 else 0), precision=prec)
else:
    g = PoseGraph.new(os.path.join(ROOT, 'tests/golden/g2o', wl + '.g2o'), precision=prec)
g.iterate_async(3); g.sync()
L = _lib.load()
L.rr_pgo_debug_flow_trace.restype = C.c_int64
L.rr_pgo_debug_flow_trace.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
lvl = 0
while True:
    nf, est = C.c_int32(), C.c_double()
    n = L.rr_pgo_debug_flow_trace(g._h, lvl, None, None, 0, C.byref(nf), C.byref(est))
    if n < 0:
        break
    tasks = np.zeros((n, 4), np.int32)
    st = np.zeros((n, 4, 4), np.uint64)
    L.rr_pgo_debug_flow_trace(g._h, lvl, tasks.ctypes.data, st.ctypes.data, n, C.byref(nf), C.byref(est))
    lvl += 1
    if only and (lvl - 1) not in only:
        continue
    t = st.astype(np.float64) * 0.01   # us
    valid = st[:, :, 0] > 0
    t0 = t[:, :, 0][valid].min()
    t = t - t0
    kind, front = tasks[:, 0] >> 24, tasks[:, 0] & 0xffffff
    end = np.where(st[:, :, 3] > 0, t[:, :, 3], 0).max()
    print(f'== flow level {lvl - 1}: {nf.value} fronts, {n} tasks ({(kind == 0).sum()} panel, {(kind == 1).sum()} update), span {end:.1f} us, cost-model critical path {est.value:.0f} us')
    # busy time: per task-wave (done - ready) against the waves the launch had
    ready, done, taken = t[:, :, 1], t[:, :, 3], t[:, :, 0]
    ok = (st[:, :, 1] > 0) & (st[:, :, 3] > 0)
    busy = (done - ready)[ok].sum()
    waitt = (ready - taken)[ok].sum()
    print(f'   wave-time: computing {busy:.0f} us, waiting for flags {waitt:.0f} us  (per wave slot of a 512-workgroup grid: {busy / 2048:.1f} / {waitt / 2048:.1f} us)')
    pan, upd = kind == 0, kind == 1
    if pan.any():
        d = (done - ready)[pan][ok[pan]]
        print(f'   PANEL waves: {len(d)}  compute mean {d.mean():.2f} us  p50 {np.median(d):.2f}  p95 {np.percentile(d, 95):.2f}')
    if upd.any():
        d = (done[:, 1] - ready[:, 1])[upd]
        x = (t[:, 1, 2] - ready[:, 1])[upd]
        print(f'   UPDATE tiles: {upd.sum()}  ready->stored mean {x.mean():.2f} us  ready->flag mean {d.mean():.2f}  p95 {np.percentile(d, 95):.2f}')
    # the chain of front 0: look waves (panel tasks with first row block 0, wave 0) and tile (0, 0)
    ch = []
    for i in range(n):
        if front[i] != 0:
            continue
        if kind[i] == 0 and tasks[i, 2] == 0:
            ch.append((tasks[i, 1], 'P', i))
        if kind[i] == 1 and tasks[i, 2] == 0 and tasks[i, 3] == 0:
            ch.append((tasks[i, 1] + 127.5, 'U', i))
    ch.sort()
    prev_done = None
    print('   chain of front 0:  kind kb   taken  pre-go    W-in    done | W-in after prev done   W-in -> done')
    rows = []
    for kb, k, i in ch:
        w = 0
        r = (k, int(kb), t[i, w, 0], t[i, w, 1], t[i, w, 2], t[i, w, 3])
        gap = (r[4] - prev_done) if prev_done is not None else 0.0
        rows.append(r + (gap, r[5] - r[4]))
        prev_done = r[5]
    for r in rows[:14] + ([('..',) * 1] if len(rows) > 14 else []) + rows[-6:] if len(rows) > 20 else rows:
        if len(r) == 1:
            print('      ...')
            continue
        print(f'      {r[0]} {r[1]:5d} {r[2]:8.2f}{r[3]:8.2f}{r[4]:8.2f}{r[5]:8.2f} | {r[6]:8.2f} {r[7]:8.2f}')
    if rows:
        gaps = np.array([r[6] for r in rows[1:]]); comp = np.array([r[7] for r in rows])
        isU = np.array([r[0] == 'U' for r in rows])
        print(f'   chain: {len(rows)} steps, sum compute {comp.sum():.0f} us (P mean {comp[~isU].mean():.2f}, U mean {comp[isU].mean() if isU.any() else 0:.2f}), '
              f'sum of gaps {gaps.sum():.0f} us (mean {gaps.mean():.2f}), ends at {rows[-1][5]:.0f} us')
