"""Pin the CPU oracle (oracle/newton_oracle.py) against golden traces recorded from the
unmodified reference (tests/golden, produced by oracle/make_golden.py) and against the
reference's own end-to-end ground truths (unit_tests.py:51,405-415; README.md:101-122)."""
import glob
import os

import numpy as np
import pytest

from oracle import newton_oracle as orc
from pyipm_amd.problems import example_problem, unit_test_x0, make_qp

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def blocks_at(prob, x, lda):
    n, me, mi = prob["nvar"], prob["neq"], prob["nineq"]
    d2L = np.array(prob["d2f"](x), dtype=np.float64)
    Je = Ji = ce = ci = None
    if me:
        d2L = d2L - prob["d2ce"](x, lda)
        Je, ce = prob["dce"](x), prob["ce"](x)
    if mi:
        d2L = d2L - prob["d2ci"](x, lda)
        Ji, ci = prob["dci"](x), prob["ci"](x)
    return d2L, Je, Ji, prob["df"](x), ce, ci


@pytest.mark.parametrize("k", range(1, 11))
def test_trace_every_iteration(k):
    d = np.load(os.path.join(GOLD, "trace_p%02d.npz" % k))
    prob = example_problem(k)
    n, me, mi = prob["nvar"], prob["neq"], prob["nineq"]
    assert (n, me, mi) == (int(d["nvar"]), int(d["neq"]), int(d["nineq"]))
    for it in range(int(d["n_iter"])):
        x, s, lda = d["it_x"][it], d["it_s"][it], d["it_lda"][it]
        d2L, Je, Ji, df, ce, ci = blocks_at(prob, x, lda)
        g = -orc.kkt_residual(df, Je, Ji, ce, ci, s, lda, d["it_mu"][it], n, me, mi)
        np.testing.assert_allclose(g, d["it_g"][it], rtol=0, atol=1e-15 * max(1.0, np.abs(g).max()))
        H = orc.kkt_matrix(d2L, Je, Ji, s, lda, n, me, mi)
        assert np.array_equal(H, d["it_H"][it]), "KKT matrix must be bit-identical"
        Hc, delta = orc.reghess(H.copy(), n, me, mi, d["it_delta_in"][it], d["it_mu_host"][it])
        assert np.array_equal(Hc, d["it_Hc"][it])
        assert delta == d["it_delta_out"][it]
        dz = orc.sym_solve(Hc, g.reshape(-1, 1)).reshape(-1)
        np.testing.assert_allclose(dz, d["it_dz_raw"][it], rtol=1e-13, atol=1e-300)
        np.testing.assert_array_equal(orc.flip_multipliers(d["it_dz_raw"][it], n, mi) if (me or mi)
                                      else d["it_dz_raw"][it], d["it_dz"][it])


@pytest.mark.parametrize("k", range(1, 11))
def test_reference_ground_truth(k):
    """The reference's own acceptance bar: ||x - x_gt||_2 <= Stol = 1e-3 (unit_tests.py:51,405-415)."""
    d = np.load(os.path.join(GOLD, "trace_p%02d.npz" % k))
    np.testing.assert_array_equal(d["x0"], unit_test_x0()[k])
    errs = [np.linalg.norm(d["x"] - gt) for gt in example_problem(k)["ground_truth"]]
    assert min(errs) <= 1e-3
    assert int(d["signal"]) in (1, 2)


def test_problem7_transcript_shape():
    """README.md:101-122: banners, then 'Converged to Ktol tolerance after ... (K total).'"""
    t = str(np.load(os.path.join(GOLD, "transcript_p07.npz"))["transcript"]).strip().splitlines()
    assert t[0] == "Searching for a feasible local minimizer using the exact Hessian."
    assert t[1] == "OUTER ITERATION 1" and t[2] == "* INNER ITERATION 1"
    assert t[-1].startswith("Converged to Ktol tolerance after 1 outer iteration and ")
    assert t[-1].endswith("total).")


def test_problem7_fixed_point():
    d = np.load(os.path.join(GOLD, "step_p07_fixed.npz"))
    prob = example_problem(7)
    d2L, Je, Ji, df, ce, ci = blocks_at(prob, d["x"], d["lda"])
    dz, delta, Hc, g = orc.newton_step(d2L, Je, Ji, df, ce, ci, d["s"], d["lda"], float(d["mu"]),
                                       3, 1, 3, delta=0.0)
    assert np.array_equal(Hc, d["Hc"]) and delta == float(d["delta_out"]) == 0.0
    np.testing.assert_allclose(dz, d["dz"], rtol=1e-13)
    # the survey's probe of the same point
    np.testing.assert_allclose(dz, [-0.5375, 0.3316, 0.0060, -0.5375, 0.3316, 0.0060, -0.5370,
                                    0.4000, 0.6665, 0.4999], atol=6e-5)
    assert int(d["neg"]) == 4


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "qp_*.npz"))))
def test_qp_golden(path):
    d = np.load(path)
    n, me, mi, seed = int(d["nvar"]), int(d["neq"]), int(d["nineq"]), int(d["seed"])
    qp = make_qp(n, me, mi, seed)
    dz, delta, Hc, g = orc.newton_step(qp["d2L"], qp["Je"], qp["Ji"], qp["df"], qp["ce"], qp["ci"],
                                       qp["s"], qp["lam"], qp["mu"], n, me, mi, delta=0.0,
                                       mu_host=float(d["mu_host"]))
    assert delta == float(d["delta_out"]) == 0.0
    np.testing.assert_allclose(g, d["g"], rtol=0, atol=1e-14 * np.abs(g).max())
    np.testing.assert_allclose(Hc.sum(axis=1), d["H_rowsum"], rtol=1e-13, atol=1e-13)
    assert np.isclose(np.trace(Hc), float(d["H_trace"]), rtol=1e-14)
    if "H" in d.files:
        assert np.array_equal(Hc, d["H"])
    np.testing.assert_allclose(dz, d["dz"], rtol=1e-11, atol=1e-13 * np.abs(dz).max())
    assert orc.inertia_from_eig(Hc)[0] == int(d["neg"]) == me + mi


def test_nonconvex_delta_loop():
    d = np.load(os.path.join(GOLD, "step_nonconvex_delta_loop.npz"))
    n, me, mi = int(d["nvar"]), int(d["neq"]), int(d["nineq"])
    H = orc.kkt_matrix(d["Q"], d["A"].T, d["G"].T, d["s"], d["lda"], n, me, mi)
    assert np.array_equal(H, d["H"])
    st = {}
    Hc, delta = orc.reghess(H.copy(), n, me, mi, 0.0, float(d["mu_host"]), stats=st)
    assert delta == float(d["delta_out"]) and delta > 1.0       # went through the x10 loop
    assert st["n_eigh"] >= 4 and st["neg"] == me + mi
    assert np.array_equal(Hc, d["Hc"])


def test_rank_deficient_delta_c():
    d = np.load(os.path.join(GOLD, "step_rankdef_delta_c.npz"))
    n, me, mi = int(d["nvar"]), int(d["neq"]), int(d["nineq"])
    H = orc.kkt_matrix(d["Q"], d["A"].T, d["G"].T, d["s"], d["lda"], n, me, mi)
    st = {}
    Hc, delta = orc.reghess(H.copy(), n, me, mi, 0.0, float(d["mu_host"]), stats=st)
    assert st["delta_c_used"] and delta == float(d["delta_out"])
    assert np.array_equal(Hc, d["Hc"])
    assert Hc[n + mi, n + mi] < 0.0


# ---------------------------------------------------------------------- L-BFGS direction
import glob  # noqa: E402


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "lbfgs_dir_*.npz"))), ids=os.path.basename)
def test_lbfgs_oracle_reproduces_reference_direction(path):
    """oracle/lbfgs_oracle.py against single calls of the UNMODIFIED reference's lbfgs_dir (pyipm.py:1184-1246)
    on QP-shaped data: unconstrained, equality-only, inequality-only, both, empty storage, rank-deficient Je."""
    from oracle import lbfgs_oracle as lo
    d = np.load(path)
    me, mi = int(d["me"]), int(d["mi"])
    reg = float(d["reg_coef"]) * float(d["eta"]) * float(d["mu"]) ** float(d["beta"])
    dz = lo.direction(d["g"], float(d["zeta"]), d["S"], d["Y"], d["SS"], d["L"], d["D"],
                      Je=d["Je"] if me else None, Ji=d["Ji"] if mi else None, s=d["s"], lda=d["lda"], reg=reg)
    tol = 1e-6 if int(d["rank_deficient"]) else 1e-11
    assert np.linalg.norm(dz - d["dz_raw"]) <= tol * np.linalg.norm(d["dz_raw"])


@pytest.mark.parametrize("k", range(1, 11))
def test_lbfgs_oracle_reproduces_reference_trace_directions(k):
    """Every direction of the reference's lbfgs=4 runs on its ten example problems, from the recorded inputs."""
    from oracle import lbfgs_oracle as lo
    from pyipm_amd.problems import example_problem
    d = np.load(os.path.join(GOLD, "lbfgs_trace_p%02d.npz" % k))
    prob = example_problem(k)
    n, me, mi = int(d["nvar"]), int(d["neq"]), int(d["nineq"])
    eps = np.finfo(float).eps
    for it in range(int(d["n_iter"])):
        m = int(d["it_m"][it])
        x = d["it_x"][it]
        Je = np.asarray(prob["dce"](x), dtype=float).reshape(n, me) if me else None
        Ji = np.asarray(prob["dci"](x), dtype=float).reshape(n, mi) if mi else None
        reg = np.sqrt(eps) * 1.0e-4 * float(d["it_mu"][it]) ** 0.4
        dz = lo.direction(d["it_g"][it], float(d["it_zeta"][it]), d["it_S"][it][:, :m], d["it_Y"][it][:, :m],
                          d["it_SS"][it][:m, :m], d["it_L"][it][:m, :m], d["it_D"][it][:m, :m], Je=Je, Ji=Ji,
                          s=d["it_s"][it], lda=d["it_lda"][it], reg=reg)
        ref = d["it_dz_raw"][it]
        assert np.linalg.norm(dz - ref) <= 1e-9 * max(np.linalg.norm(ref), 1e-300)
