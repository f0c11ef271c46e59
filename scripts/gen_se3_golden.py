#!/usr/bin/env python3
"""Generates tests/golden/se3_jacobians.json: error vector and both Jacobians of the SE(3) pose-pose factor at 48
random (Xi, Xj, Z) triples, computed INDEPENDENTLY of the oracle and of the HIP kernels, in 50-digit arithmetic.

The reference pins nothing here: its SE(3) path is never executed (`todo!()` at pose_graph_optimization.rs:241,
357,570; `linearize_pose3D_pose3D_constraint` :488-514 has no caller and SURVEY F9/F10 list its defects), so the
build defines the maths and this script is the definition made executable:

    E = Z^-1 * Xi^-1 * Xj                         (rigid transforms, quaternions in g2o order x y z w)
    e = [ t_E ; sign(w_E) * vec(q_E) ]            (g2o EdgeSE3 convention)
    X <- X * (dt, Exp(dw))  i.e.  t += R dt ,  q <- q (x) exp(dw)      (right increments, |dw| = angle)
    A = d e / d (dt_i, dw_i) ,  B = d e / d (dt_j, dw_j)  at zero increment

Nothing below is differentiated by hand: A and B are central differences of e with step 1e-18 in 50-digit mpmath
arithmetic (truncation ~1e-36, round-off ~1e-32), rounded to float64 for the fixture.

    python scripts/gen_se3_golden.py
"""
import json
import os
import random

from mpmath import mp, mpf, sqrt, sin, cos

mp.dps = 50
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def qmul(a, b):   # Hamilton product, (x, y, z, w)
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return (aw * bx + ax * bw + ay * bz - az * by,
            aw * by - ax * bz + ay * bw + az * bx,
            aw * bz + ax * by - ay * bx + az * bw,
            aw * bw - ax * bx - ay * by - az * bz)


def qconj(q):
    return (-q[0], -q[1], -q[2], q[3])


def qrot(q, v):   # q v q^-1 through the product itself (no closed form: keep it independent)
    r = qmul(qmul(q, (v[0], v[1], v[2], mpf(0))), qconj(q))
    return (r[0], r[1], r[2])


def compose(a, b):   # (ta, qa) * (tb, qb)
    rt = qrot(a[1], b[0])
    return ((a[0][0] + rt[0], a[0][1] + rt[1], a[0][2] + rt[2]), qmul(a[1], b[1]))


def inverse(a):
    qc = qconj(a[1])
    rt = qrot(qc, (-a[0][0], -a[0][1], -a[0][2]))
    return (rt, qc)


def exp_so3(w):   # unit quaternion of the rotation vector w
    th = sqrt(w[0] ** 2 + w[1] ** 2 + w[2] ** 2)
    if th == 0:
        return (mpf(0), mpf(0), mpf(0), mpf(1))
    s = sin(th / 2) / th
    return (s * w[0], s * w[1], s * w[2], cos(th / 2))


def retract(x, d):   # X * (dt, Exp(dw))
    return compose(x, ((d[0], d[1], d[2]), exp_so3((d[3], d[4], d[5]))))


def error(xi, xj, z):
    E = compose(compose(inverse(z), inverse(xi)), xj)
    s = -1 if E[1][3] < 0 else 1
    return [E[0][0], E[0][1], E[0][2], s * E[1][0], s * E[1][1], s * E[1][2]], E[1][3]


def jacobian(f, h=mpf("1e-18")):
    cols = []
    for c in range(6):
        d = [mpf(0)] * 6
        d[c] = h
        ep = f(d)
        d[c] = -h
        em = f(d)
        cols.append([(ep[r] - em[r]) / (2 * h) for r in range(6)])
    return [[cols[c][r] for c in range(6)] for r in range(6)]   # row-major 6 x 6


def random_pose(rng, spread):
    q = [mpf(rng.gauss(0, 1)) for _ in range(4)]
    n = sqrt(sum(v * v for v in q))
    return (tuple(mpf(rng.uniform(-spread, spread)) for _ in range(3)), tuple(v / n for v in q))


def main():
    rng = random.Random(20261003)
    cases = []
    while len(cases) < 48:
        xi, xj = random_pose(rng, 10.0), random_pose(rng, 10.0)
        k = len(cases)
        if k % 3 == 0:      # measurement close to the actual relative pose: small residual, like real data
            noise = ((mpf(rng.gauss(0, 0.05)),) * 1 + (mpf(rng.gauss(0, 0.05)), mpf(rng.gauss(0, 0.05))),
                     exp_so3(tuple(mpf(rng.gauss(0, 0.05)) for _ in range(3))))
            z = compose(compose(inverse(xi), xj), noise)
        else:               # unrelated measurement: large residual, both signs of w_E occur
            z = random_pose(rng, 3.0)
        e, wE = error(xi, xj, z)
        if abs(wE) < mpf("0.05"):
            continue        # e is discontinuous at w_E = 0 (the sign flip): not a differentiable point
        A = jacobian(lambda d: error(retract(xi, d), xj, z)[0])
        B = jacobian(lambda d: error(xi, retract(xj, d), z)[0])
        flat = lambda p: [float(v) for v in p[0]] + [float(v) for v in p[1]]   # noqa: E731
        cases.append({"xi": flat(xi), "xj": flat(xj), "z": flat(z), "w_E": float(wE), "e": [float(v) for v in e],
                      "A": [[float(v) for v in row] for row in A], "B": [[float(v) for v in row] for row in B]})
    out = {"generator": "scripts/gen_se3_golden.py", "digits": mp.dps, "step": "1e-18",
           "convention": "E = Z^-1 Xi^-1 Xj; e = [t_E; sign(w_E) vec(q_E)]; X <- X * (dt, Exp(dw)); quaternions x y z w",
           "cases": cases}
    path = os.path.join(ROOT, "tests", "golden", "se3_jacobians.json")
    with open(path, "w") as f:
        json.dump(out, f)
    neg = sum(1 for c in cases if c["w_E"] < 0)
    print("wrote", path, len(cases), "cases,", neg, "with w_E < 0")


if __name__ == "__main__":
    main()
