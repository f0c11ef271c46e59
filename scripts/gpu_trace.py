"""Diagnostic: order and entry times of the big-front launches (-DRRPGO_STAMPS -DRRPGO_TRACE build)."""
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
n = C.c_int32()
L.rr_pgo_debug_trace.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int64]
L.rr_pgo_debug_trace.restype = C.c_int64
buf = np.zeros(400000, dtype=np.uint64)
got = L.rr_pgo_debug_trace(g._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size)
cnt = int(buf[0]); print('trace entries', cnt, 'copied', got)
tags = buf[2:2 + 2 * cnt:2].astype(np.int64); ts = buf[3:3 + 2 * cnt:2].astype(np.int64)
order = np.argsort(ts); tags, ts = tags[order], ts[order]
half = cnt // 2   # second iteration
tags, ts = tags[half:], ts[half:]
t0 = ts[0]
names = {101: 'upd1', 102: 'upd2(next)', 103: 'upd3(rest)', 200: 'panel32', 300: 'diag32'}
lo = int(os.environ.get('FROM', '2200')); hi = lo + int(os.environ.get('COUNT', '60'))
for i in range(lo, min(hi, len(ts))):
    print(f'{i:5d} {names.get(int(tags[i]), tags[i]):12s} t={(ts[i]-t0)*0.01:10.1f} us  dt={(ts[i]-ts[i-1])*0.01:7.1f}')
