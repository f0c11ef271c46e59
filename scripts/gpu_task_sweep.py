import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustrobotics_amd import PoseGraph
for name in sys.argv[1:]:
    g = PoseGraph.new(os.path.join(ROOT, 'tests/golden/g2o', name + '.g2o'))
    st = g.stats()
    g.iterate_async(20); g.sync()
    t = time.time(); g.iterate_async(200); g.sync(); dt = time.time() - t
    print('TASK_US=%s %s: %.1f us/iter, levels %d, launches %d' % (os.environ.get('RR_PGO_TASK_US'), name, dt / 200 * 1e6, st['n_levels'], st['n_launches_per_iter']))
