"""Diagnostic: phase ticks of one workgroup (tile (2,0)) of every k_big_update launch."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustrobotics_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'rustrobotics_amd', 'librr_pgo_stamps.so')
from rustrobotics_amd import PoseGraph
w, h, e = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
g = PoseGraph.synthetic_grid(w, h, e, precision='f32')
g.iterate_async(2); g.sync()
L = _lib.load()
L.rr_pgo_debug_trace.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int64]
L.rr_pgo_debug_trace.restype = C.c_int64
buf = np.zeros(400000, dtype=np.uint64)
L.rr_pgo_debug_trace(g._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size)
cnt = int(buf[0])
tags = buf[2:2 + 2 * cnt:2].astype(np.int64); ts = buf[3:3 + 2 * cnt:2].astype(np.int64)
sel = (tags >= 600) & (tags < 700)
tags, ts = tags[sel], ts[sel]
rows = []; i = 0
while i + 3 < len(tags):
    if list(tags[i:i + 5]) == [600, 604, 601, 602, 603]: rows.append(np.diff(ts[i:i + 5])); i += 5
    else: i += 1
rows = np.array(rows)
names = ['C tile + first operand loads back', 'first chunk staged + barrier', 'k loop', 'C tile store']
print(len(rows), 'launches sampled; mean / median / max us per phase of tile (2,0):')
last = rows[-12:]
print('last 12 launches (the root level), us:'); print(np.round(last / 2400, 2))
for n, col in zip(names, rows.T): print(f'  {n:34s} {col.mean()/2400:7.2f} {np.median(col)/2400:7.2f} {col.max()/2400:7.2f}')
print('  total', rows.sum(1).mean() / 2400)
