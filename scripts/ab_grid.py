"""A/B on the lattice: GN iterations/s of one build of the library (argv: lib path, W, H, E, precision)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustrobotics_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, sys.argv[1])
from rustrobotics_amd import PoseGraph, synthetic_grid_arrays
W, H, E, prec = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
g = PoseGraph.from_arrays(*synthetic_grid_arrays(W, H, E), precision=prec)
s0 = g.state()
for rep in range(3):
    g.set_state(s0); g.iterate_async(5); g.sync()
    g.set_state(s0); t0 = time.perf_counter(); g.iterate_async(50); g.sync(); dt = time.perf_counter() - t0
    print(sys.argv[1], os.environ.get("RR_PGO_NO_CHAIN64", "-"), f'{50/dt:.1f} it/s  {dt/50*1e3:.3f} ms')
