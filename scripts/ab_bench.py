"""A/B: GN iterations/s of two builds of the library on the same box (argv: lib path, workload file)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustrobotics_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, sys.argv[1])
from rustrobotics_amd import PoseGraph
g = PoseGraph.new(os.path.join(ROOT, 'tests/golden/g2o', sys.argv[2] + '.g2o'))
s0 = g.state()
for rep in range(3):
    g.set_state(s0); g.iterate_async(20); g.sync()
    g.set_state(s0); t0 = time.perf_counter(); g.iterate_async(200); g.sync(); dt = time.perf_counter() - t0
    print(sys.argv[1], sys.argv[2], f'{200/dt:.1f} it/s')
