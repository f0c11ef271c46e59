"""Diagnostic: phase ticks (s_memtime) of the chain wave of k_big_panel32 (-DRRPGO_STAMPS -DRRPGO_TRACE build)."""
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
sel = tags >= 500
tags, ts = tags[sel], ts[sel]
# consecutive groups 500..505 belong to one launch (same wave writes them in order)
rows = []
i = 0
while i + 5 < len(tags):
    if list(tags[i:i + 6]) == [500, 501, 502, 503, 504, 505]:
        rows.append(np.diff(ts[i:i + 6])); i += 6
    else: i += 1
rows = np.array(rows)
names = ['W/acc/nxt loads issued', 'K loop', 'X = A W^T + store', 'nxt MFMAs + LDS', 'diag32 factor+invert']
print(len(rows), 'chain launches; mean ticks (2.4 GHz) per phase:')
for n, m, mx in zip(names, rows.mean(0), rows.max(0)): print(f'  {n:28s} {m:9.0f} ticks = {m/2400:6.2f} us   (max {mx/2400:.2f})')
print('  total', rows.sum(1).mean() / 2400, 'us')
