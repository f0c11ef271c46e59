import sys, os, time
sys.path.insert(0, os.getcwd())
from rustrobotics_amd import PoseGraph
p = 'tests/golden/g2o/intel.g2o'
PoseGraph.new(p).optimize(10)
for rep in range(4):
    t0 = time.perf_counter(); g = PoseGraph.new(p); t1 = time.perf_counter(); e = g.optimize(10); t2 = time.perf_counter(); del g; t3 = time.perf_counter()
    print(f'new {1e3*(t1-t0):.2f} ms, optimize {1e3*(t2-t1):.2f} ms, destroy {1e3*(t3-t2):.2f} ms, total {1e3*(t3-t0):.2f}')
