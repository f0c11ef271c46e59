import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from rustrobotics_amd import PoseGraph
p = 'tests/golden/g2o/intel.g2o'
g64 = PoseGraph.new(p); e64 = g64.optimize(30)
for prec in ('f32', 'mixed'):
    g = PoseGraph.new(p, precision=prec)
    e, n = g.optimize(30, return_norms=True)
    print(prec, 'iters', len(e) - 1, 'final chi2', e[-1], 'rel', abs(e[-1] - e64[-1]) / e64[-1], 'norms', np.array2string(np.array(n), precision=2), 'state diff', np.abs(g.state() - g64.state()).max())
