import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustrobotics_amd import PoseGraph, PoseGraphSolver
from oracle.oracle import OracleGraph
G = os.path.join(ROOT, 'tests/golden/g2o/')
names = sys.argv[1:] or ['simulation-pose-landmark', 'simulation-pose-pose', 'intel', 'input_M3500_g2o', 'dlr']
for name in names:
    p = G + name + '.g2o'
    g = PoseGraph.new(p); o = OracleGraph.load(p)
    print('==', name, g.stats())
    print(' chi2 gpu %.9f cpu %.9f' % (g.global_error(), o.global_error()))
    dxg = g.linearize_and_solve(); dxo = o.linearize_and_solve()
    print(' dx max abs diff %.3e (|dx|max %.3e)' % (np.abs(dxg - dxo).max(), np.abs(dxo).max()))
    t = time.time(); eg = g.optimize(30); dt = time.time() - t
    eo = o.optimize(30)
    print(' gpu errors', ['%.9g' % e for e in eg]); print(' cpu errors', ['%.9g' % e for e in eo])
    print(' state max abs diff %.3e ; optimize wall %.2f ms (%d its)' % (np.abs(g.state() - o.state()).max(), dt * 1e3, len(eg) - 1))
    g2 = PoseGraph.new(p)
    g2.iterate_async(5); g2.sync()
    t = time.time(); g2.iterate_async(50); g2.sync(); dt = time.time() - t
    print(' iterate: %.1f us / GN iteration' % (dt / 50 * 1e6))
    print(' profile', g2.profile(10))
