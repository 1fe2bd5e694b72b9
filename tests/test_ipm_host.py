"""CPU tests of the host IPM loop (pyipm_amd/ipm.py) with the oracle standing in for the HIP
Newton backend: the restated loop must retrace the unmodified reference iteration by iteration
(tests/golden/trace_pXX.npz), reach its ground truths (unit_tests.py:51,405-415) and print the
README transcript shape (README.md:101-122)."""
import contextlib
import io
import os

import numpy as np
import pytest

from pyipm_amd.ipm import IPM
from pyipm_amd.problems import example_problem, unit_test_x0
from backends import OracleBackend, OracleLbfgsBackend

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def make_ipm(k, backend, **kw):
    p = example_problem(k)
    return IPM(x0=unit_test_x0()[k], f=p["f"], df=p["df"], d2f=p["d2f"], ce=p["ce"], dce=p["dce"], d2ce=p["d2ce"],
               ci=p["ci"], dci=p["dci"], d2ci=p["d2ci"], backend=backend, **kw)


@pytest.mark.parametrize("k", range(1, 11))
def test_host_loop_retraces_reference(k):
    d = np.load(os.path.join(GOLD, "trace_p%02d.npz" % k))
    prob = example_problem(k)
    be = OracleBackend(prob["nvar"], prob["neq"], prob["nineq"])
    ipm = make_ipm(k, be, Ftol=1.0e-8, verbosity=-1)
    x, s, lda, fval, kkt = ipm.solve()
    assert ipm.signal == int(d["signal"])
    assert len(be.calls) == int(d["n_iter"]) == ipm.iter_count
    for it, c in enumerate(be.calls):
        np.testing.assert_allclose(c["Hc"], d["it_Hc"][it], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(c["g"], d["it_g"][it], rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(c["dz"], d["it_dz"][it], rtol=1e-6, atol=1e-10)
        assert np.isclose(c["delta"], d["it_delta_out"][it], rtol=1e-12, atol=0)
    np.testing.assert_allclose(x, d["x"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(s, d["s"], rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(lda, d["lda"], rtol=1e-6, atol=1e-9)
    assert np.isclose(float(fval), float(d["fval"]), rtol=1e-9, atol=1e-12)
    assert min(np.linalg.norm(x - gt) for gt in prob["ground_truth"]) <= 1e-3
    for i in range(4):
        np.testing.assert_allclose(np.atleast_1d(kkt[i]), d["kkt%d" % (i + 1)], rtol=1e-4, atol=1e-9)


def make_lbfgs_ipm(k, backend, memory=4, **kw):
    p = example_problem(k)
    return IPM(x0=unit_test_x0()[k], f=p["f"], df=p["df"], ce=p["ce"], dce=p["dce"], ci=p["ci"], dci=p["dci"],
               lbfgs=memory, backend=backend, **kw)


@pytest.mark.parametrize("k", range(1, 11))
def test_host_lbfgs_loop_retraces_reference(k):
    """lbfgs=4 (unit_tests.py:49): storage updates (lbfgs_update), direction inputs and the RAW direction of every
    iteration, and the final point, against the unmodified reference (tests/golden/lbfgs_trace_pXX.npz)."""
    d = np.load(os.path.join(GOLD, "lbfgs_trace_p%02d.npz" % k))
    prob = example_problem(k)
    be = OracleLbfgsBackend(prob["nvar"], prob["neq"], prob["nineq"])
    ipm = make_lbfgs_ipm(k, be, Ftol=1.0e-8, verbosity=-1)
    with np.errstate(all="ignore"):
        x, s, lda, fval, kkt = ipm.solve()
    assert ipm.signal == int(d["signal"])
    assert len(be.calls) == int(d["n_iter"]) == ipm.iter_count
    for it, c in enumerate(be.calls):
        m = int(d["it_m"][it])
        assert c["S"].shape[1] == m
        np.testing.assert_allclose(c["zeta"], d["it_zeta"][it], rtol=1e-7)
        np.testing.assert_allclose(c["g"], d["it_g"][it], rtol=1e-6, atol=1e-9)
        for key in ("S", "Y"):
            np.testing.assert_allclose(c[key], d["it_" + key][it][:, :m], rtol=1e-6, atol=1e-9)
        for key in ("SS", "L", "D"):
            np.testing.assert_allclose(c[key], d["it_" + key][it][:m, :m], rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(c["dz_raw"], d["it_dz_raw"][it], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(x, d["x"], rtol=1e-7, atol=1e-9)
    assert min(np.linalg.norm(x - gt) for gt in prob["ground_truth"]) <= 1e-3


def test_lbfgs_storage_is_bounded_and_shifts():
    """The storage reaches memory+1 pairs and then drops the oldest (pyipm.py:1300-1307); a pair of non-positive
    curvature is skipped, and more than ``lbfgs`` skips in a row reset the storage (:1359-1368)."""
    ipm = make_lbfgs_ipm(2, OracleLbfgsBackend(2, 0, 0), memory=2, verbosity=-1)
    ipm.nvar, ipm.neq, ipm.nineq = 2, 0, 0
    st = ipm.lbfgs_init()
    rng = np.random.default_rng(0)
    x = rng.standard_normal(2)
    pairs = []
    for i in range(5):
        xn = x + rng.standard_normal(2)
        dg = 2.0 * (xn - x)
        st = ipm.lbfgs_update(x, xn, dg, np.zeros(2), *st)        # g_old - g_new = dg  => curvature 2|dx|^2 > 0
        pairs.append((xn - x, dg))
        x = xn
        assert st[1].shape[1] == min(i + 1, 3)
    zeta, S, Y, SS, L, D, fail = st
    np.testing.assert_allclose(S, np.stack([q[0] for q in pairs[-3:]], axis=1))
    np.testing.assert_allclose(SS, Y.T @ Y)                        # unconstrained: "SS" is Y'Y
    np.testing.assert_allclose(L, np.triu(S.T @ Y))                # and "L" the upper-triangular S'Y
    np.testing.assert_allclose(np.diag(D), np.sum(S * Y, axis=0))
    for i in range(3):                                             # non-positive curvature: skipped, counted, then reset
        st = ipm.lbfgs_update(x, x + 1.0, np.zeros(2), np.ones(2), *st)
    assert st[1].shape[1] == 0 and st[0] == 1.0 and st[-1] == 0


def test_problem7_transcript_matches_reference():
    ref = str(np.load(os.path.join(GOLD, "transcript_p07.npz"))["transcript"])
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        make_ipm(7, OracleBackend(3, 1, 3), Ftol=1.0e-8, verbosity=1).solve()
    assert buf.getvalue() == ref


def test_step_fraction_to_boundary():
    ipm = make_ipm(5, OracleBackend(2, 0, 3), verbosity=-1)
    x = np.array([1.0, 2.0, 0.5])
    assert ipm.step(x, np.array([0.1, -0.2, 1.0])) == 1.0
    dx = np.array([-4.0, -1.0, 0.3])
    a = ipm.step(x, dx)
    exact = min(ipm.tau * x[i] / -dx[i] for i in range(3) if dx[i] < 0)
    assert 0 < a <= exact and exact - a < 1e-12


def test_validation_errors():
    p = example_problem(5)
    with pytest.raises(ValueError):
        IPM(x0=np.zeros(2), f=p["f"], df=p["df"], d2f=p["d2f"], ci=p["ci"], verbosity=-1).solve()
    with pytest.raises(NotImplementedError):
        IPM(x0=np.zeros(2), f=p["f"], df=p["df"], lbfgs=32, verbosity=-1).solve()
    with pytest.raises(AssertionError):
        IPM(x0=np.zeros(2), f=p["f"], df=p["df"], d2f=p["d2f"], mu=-1.0, verbosity=-1).solve()


def test_max_iterations_signal():
    be = OracleBackend(2, 0, 0)
    ipm = make_ipm(2, be, niter=1, miter=2, verbosity=-1)      # Rosenbrock cannot finish in 2 steps
    ipm.solve()
    assert ipm.signal == -1 and len(be.calls) == 2


@pytest.mark.parametrize("name", ["n40_me10_mi24_s11", "n64_me0_mi48_s12", "n48_me16_mi0_s13"])
def test_host_loop_retraces_reference_qp_solve(name):
    """tests/golden/qptrace_*.npz (oracle/make_golden.py --qp-trace): the unmodified reference solving a synthetic QP
    through its provider contract (pyipm.py:855-954).  The restated host loop with the oracle as Newton backend visits
    the same iterates; tests/test_gpu_qp.py asks the same of the device-resident loop."""
    from pyipm_amd.problems import make_qp, qp_callables
    d = np.load(os.path.join(GOLD, "qptrace_%s.npz" % name))
    n, me, mi, seed = int(d["n"]), int(d["me"]), int(d["mi"]), int(d["seed"])
    p = qp_callables(make_qp(n, me, mi, seed))
    be = OracleBackend(n, me, mi)
    ipm = IPM(x0=np.zeros(n), f=p["f"], df=p["df"], d2f=p["d2f"], ce=p["ce"], dce=p["dce"], d2ce=p["d2ce"],
              ci=p["ci"], dci=p["dci"], d2ci=p["d2ci"], backend=be, Ktol=float(d["Ktol"]), verbosity=-1)
    x, s, lda, fval, kkt = ipm.solve()
    assert ipm.signal == int(d["signal"]) and len(be.calls) == int(d["n_iter"])
    for it, c in enumerate(be.calls):
        np.testing.assert_allclose(c["g"], d["it_g"][it], rtol=1e-7, atol=1e-10)
        assert np.isclose(c["delta"], d["it_delta_out"][it], rtol=1e-12, atol=0)
    np.testing.assert_allclose(x, d["x"], rtol=1e-8, atol=1e-10)
    assert np.isclose(float(fval), float(d["fval"]), rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("n,m,kind", [(40, 90, "rand"), (90, 40, "rand"), (64, 64, "rand"), (120, 400, "box"), (50, 80, "dependent"),
                                      (80, 50, "dependent"), (30, 70, "illcond"), (3, 1, "rand"), (1, 4, "rand")])
def test_first_multiplier_estimate_equals_the_pseudo_inverse(n, m, kind):
    """lda0 = pinv(J) df (pyipm.py:726-730): the normal equations where J allows it, numpy's pinv elsewhere -- the same vector."""
    from pyipm_amd.ipm import pinv_apply
    rng = np.random.default_rng(n * 1000 + m)
    J = rng.standard_normal((n, m))
    if kind == "box":
        J = np.hstack([np.eye(n), -np.eye(n), rng.standard_normal((n, m - 2 * n)) / np.sqrt(n)])
    elif kind == "dependent":
        J[:, -1] = J[:, 0]
        J[-1, :] = J[0, :]
    elif kind == "illcond":
        U, _, Vt = np.linalg.svd(J, full_matrices=False)
        J = (U * np.logspace(0, -9, min(n, m))) @ Vt
    g = rng.standard_normal(n)
    ref = np.linalg.pinv(J) @ g
    assert np.linalg.norm(pinv_apply(J, g) - ref) <= 1e-10 * max(np.linalg.norm(ref), 1e-300)
