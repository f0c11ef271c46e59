"""Pins the CPU oracle (oracle/pgo_oracle.c) to the reference's OWN unit-test goldens.

Every expected value below is a literal from the reference's tests
(src/mapping/g2o.rs, src/mapping/pose_graph_optimization.rs); tolerances are the
reference's.  The data files under tests/golden/g2o/ are the reference's dataset
files (dataset/g2o/*.g2o), copied as data.
"""
import numpy as np
import pytest

from oracle.oracle import OracleGraph, GAUSS_NEWTON
from conftest import g2o_path


# g2o.rs:149-175  `from_g2o`
@pytest.mark.parametrize("name,nodes,edges,dim", [
    ("simulation-pose-pose", 400, 1773, 1200),
    ("simulation-pose-landmark", 77, 297, 195),
    ("intel", 1728, 4830, 5184),
    ("dlr", 3873, 17605, 11043),
])
def test_from_g2o(name, nodes, edges, dim):
    g = OracleGraph.load(g2o_path(name))
    assert (g.num_nodes, g.num_edges, g.dim) == (nodes, edges, dim)


# pose_graph_optimization.rs:580-598  `initial_global_error`
@pytest.mark.parametrize("name,expected,eps", [
    ("simulation-pose-pose", 138862234.0, 10.0),
    ("simulation-pose-landmark", 3030.0, 1.0),
    ("intel", 1795139.0, 1e-2),
    ("dlr", 369655336.0, 10.0),
])
def test_initial_global_error(name, expected, eps):
    g = OracleGraph.load(g2o_path(name))
    assert abs(g.global_error() - expected) <= eps


# pose_graph_optimization.rs:600-631  `final_global_error` (GN, optimize(100,..))
@pytest.mark.parametrize("name,expected", [
    ("simulation-pose-pose", 8269.0),
    ("simulation-pose-landmark", 474.0),
    ("intel", 360.0),
    ("dlr", 56860.0),
])
def test_final_global_error(name, expected):
    g = OracleGraph.load(g2o_path(name))
    errors = g.optimize(100, GAUSS_NEWTON)
    assert abs(errors[-1] - expected) <= 1.0


# pose_graph_optimization.rs:633-690  `linearize_pose_pose_constraint_correct`
def test_linearize_pose_pose_constraint_correct():
    g = OracleGraph.load(g2o_path("simulation-pose-landmark"))
    A, B, e = g.linearize_edge(0)
    np.testing.assert_allclose(A, [[0.0, 1.0, 0.113], [-1.0, 0.0, 0.024], [0.0, 0.0, -1.0]], atol=1e-3)
    np.testing.assert_allclose(B, [[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]], atol=1e-3)
    np.testing.assert_allclose(e, np.zeros(3), atol=1e-3)
    A, B, e = g.linearize_edge(10)
    np.testing.assert_allclose(A, [[0.037, 0.999, 0.138], [-0.999, 0.037, -0.982], [0.0, 0.0, -1.0]], atol=1e-3)
    np.testing.assert_allclose(B, [[-0.037, -0.999, 0.0], [0.999, -0.037, 0.0], [0.0, 0.0, 1.0]], atol=1e-3)
    np.testing.assert_allclose(e, np.zeros(3), atol=1e-3)


# pose_graph_optimization.rs:692-722  `linearize_pose_landmark_constraint_correct`
def test_linearize_pose_landmark_constraint_correct():
    g = OracleGraph.load(g2o_path("simulation-pose-landmark"))
    A, B, e = g.linearize_edge(1)
    np.testing.assert_allclose(A, [[0.0, 1.0, 0.358], [-1.0, 0.0, -0.051]], atol=1e-3)
    np.testing.assert_allclose(B, [[0.0, -1.0], [1.0, 0.0]], atol=1e-3)
    np.testing.assert_allclose(e, np.zeros(2), atol=1e-3)


# pose_graph_optimization.rs:724-739  `linearize_and_solve_correct`
def test_linearize_and_solve_correct():
    g = OracleGraph.load(g2o_path("simulation-pose-landmark"))
    dx = g.linearize_and_solve()
    expected = [1.68518905e-01, 5.74311089e-01, -5.08805168e-02, -3.67482151e-02, 8.89458085e-01]
    np.testing.assert_allclose(dx[:5], expected, atol=1e-3)


# SE(3): nothing in the reference pins it (its SE(3) path is todo!(), :241,357,570; :488-514 is never called), so
# the build-defined factor is pinned to an independent 50-digit evaluation instead (scripts/gen_se3_golden.py).
def test_se3_factor_matches_the_50_digit_fixture():
    import json
    import os
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "se3_jacobians.json")))
    assert len(fx["cases"]) == 48 and sum(c["w_E"] < 0 for c in fx["cases"]) >= 8
    info = np.eye(6)[np.triu_indices(6)]
    for c in fx["cases"]:
        g = OracleGraph.from_arrays([2, 2], np.array(c["xi"] + c["xj"]), [2], [0], [1], np.array(c["z"]), info)
        A, B, e = g.linearize_edge(0)
        np.testing.assert_allclose(e, c["e"], rtol=0, atol=1e-13)
        np.testing.assert_allclose(A, c["A"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(B, c["B"], rtol=0, atol=1e-12)
        assert abs(g.global_error() - np.dot(c["e"], c["e"])) <= 1e-12 * max(1.0, np.dot(c["e"], c["e"]))
