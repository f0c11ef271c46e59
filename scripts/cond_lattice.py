"""cond(H) of the 400 x 250 / 1M-edge lattice with the reference prior (CPU, ~4 minutes: SuperLU shift-invert + eigsh): the error budget
behind the rtol of test_config4_lattice_f64_matches_the_oracle_fixture.  Result (r05): lambda_max 1.008e7, lambda_min 6.82e-6, cond 1.48e12."""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scipy.sparse as sp, scipy.sparse.linalg as spl
from rustrobotics_amd import synthetic_grid_arrays
from oracle.oracle import OracleGraph
arrays = synthetic_grid_arrays(400, 250, 1000000)
o = OracleGraph.from_arrays(*arrays)
colptr, rowidx, vals, b = o.build_system()
n = o.dim
L = sp.csc_matrix((vals, rowidx, colptr), shape=(n, n))
H = (L + sp.tril(L, -1).T).tocsc()
print('nnz', H.nnz, flush=True)
lmax = spl.eigsh(H, k=1, which='LA', return_eigenvectors=False, tol=1e-6)[0]
print('lambda_max', lmax, flush=True)
t0 = time.time()
lu = spl.splu(H, permc_spec='MMD_AT_PLUS_A', options=dict(SymmetricMode=True))
print('splu', time.time() - t0, flush=True)
op = spl.LinearOperator(H.shape, matvec=lu.solve, dtype=np.float64)
imax = spl.eigsh(op, k=1, which='LA', return_eigenvectors=False, tol=1e-6)[0]
print('lambda_min', 1.0 / imax, 'cond', lmax * imax, flush=True)
