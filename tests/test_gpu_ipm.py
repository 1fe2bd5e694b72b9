"""End-to-end: pyipm_amd.IPM.solve() with the HIP Newton backend on the reference's ten example
problems (seed-42 starting points of unit_tests.py).  Bars: the reference's own acceptance test
||x - x_gt|| <= 1e-3 (unit_tests.py:51,405-415), plus agreement with the unmodified reference's
run (tests/golden/trace_pXX.npz): same signal, same iteration count, same delta decisions, final
x / s / lda to tight tolerance, and the README transcript for problem 7 (config 1 of BASELINE.json)."""
import contextlib
import io
import os

import numpy as np
import pytest

from pyipm_amd.problems import example_problem, unit_test_x0, make_qp, qp_callables

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def make_ipm(k, **kw):
    from pyipm_amd.ipm import IPM
    p = example_problem(k)
    return IPM(x0=unit_test_x0()[k], f=p["f"], df=p["df"], d2f=p["d2f"], ce=p["ce"], dce=p["dce"], d2ce=p["d2ce"],
               ci=p["ci"], dci=p["dci"], d2ci=p["d2ci"], **kw)


@pytest.mark.parametrize("k", range(1, 11))
def test_solve_matches_reference_run(k):
    d = np.load(os.path.join(GOLD, "trace_p%02d.npz" % k))
    prob = example_problem(k)
    ipm = make_ipm(k, Ftol=1.0e-8, verbosity=-1)
    x, s, lda, fval, kkt = ipm.solve()
    assert min(np.linalg.norm(x - gt) for gt in prob["ground_truth"]) <= 1e-3      # the reference's own bar
    assert ipm.signal == int(d["signal"])
    assert ipm.iter_count == int(d["n_iter"])
    np.testing.assert_allclose(x, d["x"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(s, d["s"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(lda, d["lda"], rtol=1e-5, atol=1e-8)
    assert np.isclose(float(fval), float(d["fval"]), rtol=1e-8, atol=1e-11)
    assert np.isclose(ipm.delta, float(d["it_delta_out"][-1]), rtol=1e-12, atol=0) if int(d["n_iter"]) else True


def test_problem7_transcript_on_gpu():
    ref = str(np.load(os.path.join(GOLD, "transcript_p07.npz"))["transcript"])
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        make_ipm(7, Ftol=1.0e-8, verbosity=1).solve()
    assert buf.getvalue() == ref


def test_nonconvex_start_triggers_device_inertia_correction():
    """Problem 4 from a point where the Lagrangian Hessian is indefinite: the backend must shift delta
    exactly as the reference's eigen-based reghess did (same delta sequence recorded in the trace)."""
    d = np.load(os.path.join(GOLD, "trace_p04.npz"))
    ipm = make_ipm(4, Ftol=1.0e-8, verbosity=-1)
    ipm.solve()
    assert ipm.backend.n_factor >= int(d["n_iter"])
    assert np.isclose(ipm.delta, float(d["it_delta_out"][-1]), rtol=1e-12, atol=0)


def test_qp_solve_medium():
    """A 300-variable QP through the whole loop; optimality checked by the KKT residuals."""
    from pyipm_amd.ipm import IPM
    qp = make_qp(300, 40, 120, seed=12)
    p = qp_callables(qp)
    x0 = np.zeros(300)
    ipm = IPM(x0=x0, f=p["f"], df=p["df"], d2f=p["d2f"], ce=p["ce"], dce=p["dce"], d2ce=p["d2ce"], ci=p["ci"],
              dci=p["dci"], d2ci=p["d2ci"], verbosity=-1, niter=20, Ktol=1e-6)
    x, s, lda, fval, kkt = ipm.solve()
    assert ipm.signal == 1
    assert all(np.linalg.norm(k) <= 1e-6 for k in kkt)
    assert np.all(s > 0) and np.all(lda[40:] > -1e-12)
    assert np.linalg.norm(qp["A"] @ x - qp["b"]) <= 1e-6 and np.min(qp["G"] @ x - qp["h"]) >= -1e-6


def test_cli_problem7(capsys):
    """BASELINE.json configs[0]: the `python pyipm.py 7` plumbing run, here `python -m pyipm_amd 7 <seed>`."""
    from pyipm_amd.__main__ import main
    assert main(["7", "3"]) == 0
    out = capsys.readouterr().out.splitlines()
    assert out[0] == "Searching for a feasible local minimizer using the exact Hessian."
    assert out[1] == "OUTER ITERATION 1" and out[2] == "* INNER ITERATION 1"
    conv = [l for l in out if l.startswith("Converged to Ktol tolerance")]
    assert conv and conv[0].endswith("total).")
    sol = [l for l in out if l.startswith("Solver solution")][0]
    vals = np.array([float(v) for v in sol.split("[")[1].rstrip("]").split(",")])
    assert np.linalg.norm(vals - 1.0 / 3.0) <= 1e-3


def test_device_step_lengths_match_host_search():
    """closed-form fraction-to-the-boundary on the device vs the reference's golden-section
    search restated on the host (pyipm.py:1408-1436): same alpha to 1e-15 on every Newton system of a trace."""
    from pyipm_amd.newton import NewtonCore
    from pyipm_amd.ipm import IPM
    for k in (5, 6, 7, 9, 10):
        d = np.load(os.path.join(GOLD, "trace_p%02d.npz" % k))
        prob = example_problem(k)
        n, me, mi = prob["nvar"], prob["neq"], prob["nineq"]
        host = IPM(x0=unit_test_x0()[k], f=prob["f"], df=prob["df"], d2f=prob["d2f"], backend=object(), verbosity=-1)
        core = NewtonCore(n, me, mi, device=0)
        for it in range(int(d["n_iter"])):
            x, s, lda = d["it_x"][it], d["it_s"][it], d["it_lda"][it]
            d2L = np.array(prob["d2f"](x), dtype=np.float64)
            Je = ce = None
            if me:
                d2L = d2L - prob["d2ce"](x, lda); Je, ce = prob["dce"](x), prob["ce"](x)
            d2L = d2L - prob["d2ci"](x, lda)
            core.stage_blocks(d2L, Je, prob["dci"](x))
            core.stage_vectors(prob["df"](x), ce, prob["ci"](x), s, lda, mu=float(d["it_mu"][it]))
            delta = float(d["it_delta_out"][it]) if not np.array_equal(d["it_Hc"][it], d["it_H"][it]) else 0.0
            dz, _ = core.step(delta, 0.0)
            dz = dz.cpu().numpy()
            a_s, a_l = core.step_lengths(0.995)
            h_s = host.step(s, dz[n:n + mi])
            h_l = host.step(lda[me:], dz[n + mi + me:])
            assert abs(a_s - h_s) <= 1e-15 * max(1.0, h_s) + 4e-16 and abs(a_l - h_l) <= 1e-15 * max(1.0, h_l) + 4e-16
            ds, dl = dz[n:n + mi], dz[n + mi + me:]
            exact_s = min([1.0] + [-0.995 * s[i] / ds[i] for i in range(mi) if ds[i] < 0])
            assert a_s == exact_s


@pytest.mark.parametrize("k", [5, 7, 10])
def test_solve_with_device_step_lengths(k):
    d = np.load(os.path.join(GOLD, "trace_p%02d.npz" % k))
    ipm = make_ipm(k, Ftol=1.0e-8, verbosity=-1, device_step=True)
    x, s, lda, fval, kkt = ipm.solve()
    assert ipm.signal == int(d["signal"]) and ipm.iter_count == int(d["n_iter"])
    np.testing.assert_allclose(x, d["x"], rtol=1e-7, atol=1e-9)


def test_backend_follows_the_problem_shape_and_condensed_fallback_lasts_one_solve():
    """compile() kept the backend of the first problem when force_recompile brought another shape,
    and one condensed-form fallback disabled the condensed option for the backend's lifetime."""
    from pyipm_amd.ipm import IPM
    qa, qb = make_qp(40, 10, 20, 1), make_qp(64, 0, 48, 2)
    pa, pb = qp_callables(qa), qp_callables(qb)
    ipm = IPM(x0=np.zeros(40), f=pa["f"], df=pa["df"], d2f=pa["d2f"], ce=pa["ce"], dce=pa["dce"], d2ce=pa["d2ce"],
              ci=pa["ci"], dci=pa["dci"], d2ci=pa["d2ci"], verbosity=-1, condensed=True, Ktol=1e-8, niter=30, miter=30)
    xa = ipm.solve()[0]
    assert ipm.signal == 1 and ipm.backend.shape() == (40, 10, 20)
    first = ipm.backend
    # pretend the condensed form failed once in that solve: the next solve must try it again
    first.condensed_on = False
    first.core.set_option("condensed", 0)
    xa2 = ipm.solve()[0]
    assert first.condensed_on and ipm.backend is first
    np.testing.assert_allclose(xa2, xa, rtol=1e-8, atol=1e-10)
    # another problem through the same object
    for k in ("f", "df", "d2f", "ce", "dce", "d2ce", "ci", "dci", "d2ci"):
        setattr(ipm, k, pb[k])
    xb = ipm.solve(x0=np.zeros(64), force_recompile=True)[0]
    assert ipm.backend is not first and ipm.backend.shape() == (64, 0, 48) and ipm.signal == 1
    ref = IPM(x0=np.zeros(64), f=pb["f"], df=pb["df"], d2f=pb["d2f"], ci=pb["ci"], dci=pb["dci"], d2ci=pb["d2ci"],
              verbosity=-1, Ktol=1e-8, niter=30, miter=30).solve()[0]
    np.testing.assert_allclose(xb, ref, rtol=1e-6, atol=1e-8)
