import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustrobotics_amd import PoseGraph, synthetic_grid_arrays
W, H, E, prec, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
g = PoseGraph.from_arrays(*synthetic_grid_arrays(W, H, E), precision=prec)
g.iterate_async(iters); g.sync()
