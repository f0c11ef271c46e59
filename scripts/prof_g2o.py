"""Diagnostic: a few GN iterations of a g2o dataset with plain launches, for `rocprofv3 --kernel-trace --stats`."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustrobotics_amd import PoseGraph
g = PoseGraph.new(os.path.join(ROOT, 'tests/golden/g2o', sys.argv[1] + '.g2o'))
s0 = g.state()
for rep in range(int(sys.argv[2])):
    g.set_state(s0); g.iterate_async(1); g.sync()
