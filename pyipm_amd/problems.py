"""Example problems and the synthetic dense-QP generator.

* ``example_problem(k)`` — the ten toy problems of the reference's ``main()``
  (``/root/reference/pyipm.py:1919-2131``; the same ten appear as p1..p10 in
  ``unit_tests.py:96-237``), restated as plain NumPy callables with hand-derived
  first and second derivatives.  Signatures follow the reference's
  "precompiled" input convention (pyipm.py:85-146, 512-562):
  ``f(x)``, ``df(x)``, ``d2f(x)``, ``ce(x)``, ``dce(x) -> (n, me)``,
  ``d2ce(x, lda)``, ``ci(x)``, ``dci(x) -> (n, mi)``, ``d2ci(x, lda)`` where the
  constraint Hessians are Hessians of ``sum(c * lda_block)`` (pyipm.py:491-493,
  505-507) and take the FULL multiplier vector ``lda = [lda_e ; lda_i]``.
* ``unit_test_x0()`` — the seed-42 starting points of ``unit_tests.py:8,103-234``.
* ``make_qp(n, me, mi, seed)`` — the convex QP generator of SURVEY.md section 8(d)
  used by the bench and the parity tests.
"""
from __future__ import annotations

import numpy as np

EPS = float(np.finfo(np.float64).eps)


def example_problem(k: int) -> dict:
    """Return problem ``k`` (1..10) as a dict of callables + metadata."""
    p = {"f": None, "df": None, "d2f": None, "ce": None, "dce": None, "d2ce": None,
         "ci": None, "dci": None, "d2ci": None, "neq": 0, "nineq": 0}
    A = np.array
    if k == 1:      # pyipm.py:1920-1935   min x^2 - 4x + y^2 - y - xy
        p.update(nvar=2, ground_truth=[A([3.0, 2.0])],
                 f=lambda x: x[0] ** 2 - 4 * x[0] + x[1] ** 2 - x[1] - x[0] * x[1],
                 df=lambda x: A([2 * x[0] - 4 - x[1], 2 * x[1] - 1 - x[0]]),
                 d2f=lambda x: A([[2.0, -1.0], [-1.0, 2.0]]))
    elif k == 2:    # pyipm.py:1936-1952   Rosenbrock
        p.update(nvar=2, ground_truth=[A([1.0, 1.0])],
                 f=lambda x: 100 * (x[1] - x[0] ** 2) ** 2 + (1 - x[0]) ** 2,
                 df=lambda x: A([-400 * x[0] * (x[1] - x[0] ** 2) - 2 * (1 - x[0]),
                                 200 * (x[1] - x[0] ** 2)]),
                 d2f=lambda x: A([[1200 * x[0] ** 2 - 400 * x[1] + 2, -400 * x[0]],
                                  [-400 * x[0], 200.0]]))
    elif k == 3:    # pyipm.py:1953-1970   max x + y  s.t. x^2 + y^2 = 1
        r = np.sqrt(2.0) / 2.0
        p.update(nvar=2, neq=1, ground_truth=[A([r, r])],
                 f=lambda x: -np.sum(x),
                 df=lambda x: A([-1.0, -1.0]),
                 d2f=lambda x: np.zeros((2, 2)),
                 ce=lambda x: A([np.sum(x ** 2) - 1.0]),
                 dce=lambda x: A([[2 * x[0]], [2 * x[1]]]),
                 d2ce=lambda x, lda: 2.0 * lda[0] * np.eye(2))
    elif k == 4:    # pyipm.py:1971-1994   max x^2 y  s.t. x^2 + y^2 = 3
        p.update(nvar=2, neq=1,
                 ground_truth=[A([np.sqrt(2.0), 1.0]), A([-np.sqrt(2.0), 1.0]), A([0.0, -np.sqrt(3.0)])],
                 f=lambda x: -(x[0] ** 2) * x[1],
                 df=lambda x: A([-2 * x[0] * x[1], -x[0] ** 2]),
                 d2f=lambda x: A([[-2 * x[1], -2 * x[0]], [-2 * x[0], 0.0]]),
                 ce=lambda x: A([np.sum(x ** 2) - 3.0]),
                 dce=lambda x: A([[2 * x[0]], [2 * x[1]]]),
                 d2ce=lambda x, lda: 2.0 * lda[0] * np.eye(2))
    elif k == 5:    # pyipm.py:1995-2018   QP with three linear inequalities
        p.update(nvar=2, nineq=3, ground_truth=[A([4.0, 3.0])],
                 f=lambda x: x[0] ** 2 + 2 * x[1] ** 2 + 2 * x[0] + 8 * x[1],
                 df=lambda x: A([2 * x[0] + 2, 4 * x[1] + 8]),
                 d2f=lambda x: A([[2.0, 0.0], [0.0, 4.0]]),
                 ci=lambda x: A([x[0] + 2 * x[1] - 10.0, x[0], x[1]]),
                 dci=lambda x: A([[1.0, 1.0, 0.0], [2.0, 0.0, 1.0]]),
                 d2ci=lambda x, lda: np.zeros((2, 2)))
    elif k == 6:    # pyipm.py:2019-2042   max-entropy die
        p.update(nvar=6, neq=1, nineq=6, ground_truth=[np.full(6, 1.0 / 6.0)],
                 f=lambda x: np.sum(x * np.log(x + EPS)),
                 df=lambda x: np.log(x + EPS) + x / (x + EPS),
                 d2f=lambda x: np.diag(1.0 / (x + EPS) + EPS / (x + EPS) ** 2),
                 ce=lambda x: A([np.sum(x) - 1.0]),
                 dce=lambda x: np.ones((6, 1)),
                 d2ce=lambda x, lda: np.zeros((6, 6)),
                 ci=lambda x: 1.0 * x,
                 dci=lambda x: np.eye(6),
                 d2ci=lambda x, lda: np.zeros((6, 6)))
    elif k == 7:    # pyipm.py:2043-2064   max xyz on the simplex (BASELINE config 1)
        p.update(nvar=3, neq=1, nineq=3, ground_truth=[np.full(3, 1.0 / 3.0)],
                 f=lambda x: -x[0] * x[1] * x[2],
                 df=lambda x: A([-x[1] * x[2], -x[0] * x[2], -x[0] * x[1]]),
                 d2f=lambda x: A([[0.0, -x[2], -x[1]], [-x[2], 0.0, -x[0]], [-x[1], -x[0], 0.0]]),
                 ce=lambda x: A([np.sum(x) - 1.0]),
                 dce=lambda x: np.ones((3, 1)),
                 d2ce=lambda x, lda: np.zeros((3, 3)),
                 ci=lambda x: 1.0 * x,
                 dci=lambda x: np.eye(3),
                 d2ci=lambda x, lda: np.zeros((3, 3)))
    elif k == 8:    # pyipm.py:2065-2087   linear cost, one linear + one quadratic equality
        s13 = np.sqrt(13.0)
        p.update(nvar=3, neq=2, ground_truth=[A([2.0 / s13, -3.0 / s13, -2.0 + 7.0 / s13])],
                 f=lambda x: 4.0 * x[1] - 2.0 * x[2],
                 df=lambda x: A([0.0, 4.0, -2.0]),
                 d2f=lambda x: np.zeros((3, 3)),
                 ce=lambda x: A([2 * x[0] - x[1] - x[2] - 2.0, x[0] ** 2 + x[1] ** 2 - 1.0]),
                 dce=lambda x: A([[2.0, 2 * x[0]], [-1.0, 2 * x[1]], [-1.0, 0.0]]),
                 d2ce=lambda x, lda: lda[1] * np.diag([2.0, 2.0, 0.0]))
    elif k == 9:    # pyipm.py:2088-2108   QP with two linear inequalities
        p.update(nvar=2, nineq=2, ground_truth=[A([5.0 / 3.0, 1.0 / 3.0])],
                 f=lambda x: (x[0] - 2.0) ** 2 + 2.0 * (x[1] - 1.0) ** 2,
                 df=lambda x: A([2 * (x[0] - 2.0), 4 * (x[1] - 1.0)]),
                 d2f=lambda x: A([[2.0, 0.0], [0.0, 4.0]]),
                 ci=lambda x: A([-x[0] - 4 * x[1] + 3.0, x[0] - x[1]]),
                 dci=lambda x: A([[-1.0, 1.0], [-4.0, -1.0]]),
                 d2ci=lambda x, lda: np.zeros((2, 2)))
    elif k == 10:   # pyipm.py:2109-2131   quadratic cost, linear equality, quadratic inequality
        p.update(nvar=3, neq=1, nineq=1, ground_truth=[A([0.12288, -1.1078, 0.015100])],
                 f=lambda x: (x[0] - 1.0) ** 2 + 2 * (x[1] + 2.0) ** 2 + 3 * (x[2] + 3.0) ** 2,
                 df=lambda x: A([2 * (x[0] - 1.0), 4 * (x[1] + 2.0), 6 * (x[2] + 3.0)]),
                 d2f=lambda x: np.diag([2.0, 4.0, 6.0]),
                 ce=lambda x: A([x[2] - x[1] - x[0] - 1.0]),
                 dce=lambda x: A([[-1.0], [-1.0], [1.0]]),
                 d2ce=lambda x, lda: np.zeros((3, 3)),
                 ci=lambda x: A([x[2] - x[0] ** 2]),
                 dci=lambda x: A([[-2 * x[0]], [0.0], [1.0]]),
                 d2ci=lambda x, lda: lda[1] * np.diag([-2.0, 0.0, 0.0]))
    else:
        raise ValueError("problem number must be 1..10")
    p["number"] = k
    return p


def unit_test_x0() -> dict:
    """Starting points drawn exactly as ``unit_tests.py`` draws them: legacy
    ``np.random.seed(42)`` (unit_tests.py:8), one draw per problem in file order
    p1..p10 (:103,118,130,142,163,178,190,206,221,234) — the commented-out problems
    still consume their draws."""
    rs = np.random.RandomState(42)
    x0 = {}
    for k in (1, 2, 3, 4, 5):
        x0[k] = rs.randn(2)
    x0[6] = rs.rand(6)
    x0[7] = rs.randn(3)
    x0[8] = rs.randn(3)
    x0[9] = rs.randn(2)
    x0[10] = rs.randn(3)
    return x0


def make_qp(n: int, me: int, mi: int, seed: int = 0, dtype=np.float64) -> dict:
    """Synthetic convex QP of SURVEY.md section 8(d):

        min 1/2 x'Qx + c'x   s.t.  Ax = b,  Gx - h >= 0

    Q = M M'/n + I,  A, G ~ N(0,1)/sqrt(n),  c ~ N(0,1);  start x = 0,
    s ~ U(0.5,2), lam_i ~ U(0.5,2), lam_e ~ N(0,1);  h = Gx - s - N(0,0.1)
    (so ci - s is a small non-zero residual),  b = Ax - N(0,0.1);  mu = 0.2.

    Returns the derivative blocks in the reference's layout: d2L = Q (n x n),
    Je = A' (n x me), Ji = G' (n x mi), df = Qx + c, ce = Ax - b, ci = Gx - h.
    """
    rng = np.random.default_rng(seed)
    M = rng.standard_normal((n, n))
    Q = M @ M.T / n + np.eye(n)
    del M
    A = rng.standard_normal((me, n)) / np.sqrt(n)
    G = rng.standard_normal((mi, n)) / np.sqrt(n)
    c = rng.standard_normal(n)
    x = np.zeros(n)
    s = rng.uniform(0.5, 2.0, mi)
    lam_i = rng.uniform(0.5, 2.0, mi)
    lam_e = rng.standard_normal(me)
    h = G @ x - s - 0.1 * rng.standard_normal(mi)
    b = A @ x - 0.1 * rng.standard_normal(me)
    lam = np.concatenate([lam_e, lam_i])
    return {
        "n": n, "me": me, "mi": mi, "seed": seed, "mu": 0.2,
        "Q": Q.astype(dtype), "c": c, "A": A, "b": b, "G": G, "h": h,
        "x": x, "s": s, "lam": lam,
        "d2L": Q.astype(dtype),
        "Je": np.ascontiguousarray(A.T), "Ji": np.ascontiguousarray(G.T),
        "df": Q @ x + c, "ce": A @ x - b, "ci": G @ x - h,
    }


def qp_callables(qp: dict) -> dict:
    """The QP as reference-convention callables (for driving ``IPM.solve``)."""
    Q, c, A, b, G, h = qp["Q"], qp["c"], qp["A"], qp["b"], qp["G"], qp["h"]
    n, me, mi = qp["n"], qp["me"], qp["mi"]
    p = {"nvar": n, "neq": me, "nineq": mi,
         "f": lambda x: 0.5 * x @ (Q @ x) + c @ x,
         "df": lambda x: Q @ x + c,
         "d2f": lambda x: Q,
         "ce": None, "dce": None, "d2ce": None, "ci": None, "dci": None, "d2ci": None}
    if me:
        JeT = np.ascontiguousarray(A.T)
        p.update(ce=lambda x: A @ x - b, dce=lambda x: JeT, d2ce=lambda x, lda: np.zeros((n, n)))
    if mi:
        JiT = np.ascontiguousarray(G.T)
        p.update(ci=lambda x: G @ x - h, dci=lambda x: JiT, d2ci=lambda x, lda: np.zeros((n, n)))
    return p
