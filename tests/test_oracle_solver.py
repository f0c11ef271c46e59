"""Independent check of the oracle's own sparse solver and of its LM / update restatements
(the parts no reference golden pins): SciPy SuperLU on the oracle's assembled system."""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from oracle.oracle import OracleGraph, LEVENBERG_MARQUARDT
from conftest import g2o_path


def _full(n, colptr, rowidx, vals):
    L = sp.csc_matrix((vals, rowidx, colptr), shape=(n, n))
    return (L + sp.tril(L, -1).T).tocsc()


def test_oracle_cholesky_matches_superlu():
    for name in ("simulation-pose-landmark", "simulation-pose-pose", "intel"):
        g = OracleGraph.load(g2o_path(name))
        colptr, rowidx, vals, b = g.build_system()
        H = _full(g.dim, colptr, rowidx, vals)
        dx_lu = spla.splu(H).solve(b)
        dx = g.linearize_and_solve()
        assert np.abs(dx - dx_lu).max() <= 1e-7 * max(1.0, np.abs(dx_lu).max()), name


def test_oracle_system_structure_intel():
    """nnz(H) after summing COO duplicates = 9 (N + 2E) -> lower triangle 6N + 9E."""
    g = OracleGraph.load(g2o_path("intel"))
    colptr, rowidx, vals, b = g.build_system()
    assert len(vals) == 6 * 1728 + 9 * 4830
    # prior: +1e7 on the diagonal of the from-node of the first EDGE_SE2 (node id 2), :330-336
    ef, _ = g.edge_endpoints()
    off = g.node_offsets()[ef[0]]
    assert g.node_ids()[ef[0]] == 2
    H = _full(g.dim, colptr, rowidx, vals)
    assert all(H[off + i, off + i] > 1e7 for i in range(3))
    # LM adds lambda on every diagonal, :362-366
    c2, r2, v2, _ = g.build_system(0.5, True)
    H2 = _full(g.dim, c2, r2, v2)
    assert np.allclose((H2 - H).diagonal(), 0.5)


def test_oracle_lm_quirks():
    """LM (:275-286): rejected steps are undone but their error is still recorded; GN and LM
    agree on the final chi2 on an easy graph."""
    g = OracleGraph.load(g2o_path("simulation-pose-landmark"))
    e_lm = g.optimize(40, LEVENBERG_MARQUARDT)
    g2 = OracleGraph.load(g2o_path("simulation-pose-landmark"))
    e_gn = g2.optimize(40)
    assert abs(e_lm[-1] - e_gn[-1]) < 1e-3
    assert e_lm[0] == e_gn[0]


def test_oracle_update_is_complex_product():
    """update_nodes (:229-245): t += dx.xy, R <- R * (cos, sin)(dtheta), no renormalisation."""
    g = OracleGraph.load(g2o_path("simulation-pose-landmark"))
    kinds, offs = g.node_kinds(), g.node_offsets()
    node = int(np.where(kinds == 0)[0][3])
    before = g.se2_raw(node)
    dx = np.zeros(g.dim)
    dx[offs[node]:offs[node] + 3] = [0.25, -0.5, 0.3]
    g.update_nodes(dx)
    after = g.se2_raw(node)
    c, s = np.cos(0.3), np.sin(0.3)
    expect = [before[0] + 0.25, before[1] - 0.5, before[2] * c - before[3] * s, before[2] * s + before[3] * c]
    assert np.allclose(after, expect, atol=1e-15)
    g.update_nodes(dx, -1.0)
    assert np.allclose(g.se2_raw(node), before, atol=1e-15)


def test_se3_datasets_load_and_parking_garage_converges():
    """The reference's SE(3) files beyond sphere2500, oracle only: vertex/edge counts and
    the Gauss-Newton minimum of parking-garage.g2o under the build-defined SE(3) factor (unpinned in the
    reference, whose SE(3) path is todo!())."""
    from oracle.oracle import OracleGraph
    from conftest import g2o_path
    t = OracleGraph.load(g2o_path("torus3D"))
    assert (t.num_nodes, t.num_edges, t.dim) == (5000, 9048, 30000)
    assert abs(t.global_error() - 2946826.5386780924) < 1e-3
    p = OracleGraph.load(g2o_path("parking-garage"))
    assert (p.num_nodes, p.num_edges, p.dim) == (1661, 6275, 9966)
    e = p.optimize(8)
    assert abs(e[0] - 16720.018170518) < 1e-6 and abs(e[-1] - 1.238691) < 1e-5
