"""SURVEY.md section 8(f) rank 1, second half: the merit-function pieces of the line search as device reductions of the
library (include/pyipm_newton.h: pyipm_newton_merit_info / _dots / _merit_ray; kernels_merit.hpp) against the HOST merit
function of pyipm_amd/ipm.py -- itself a restatement of pyipm.py:670-721 (phi, dphi), :1727-1735 (nu threshold), :958-991
(KKT report), :1804-1814 (barrier sums) -- on every iterate of the reference's own traces (tests/golden/trace_p*.npz:
nonlinear problems; qptrace_*.npz: QPs solved by the unmodified reference)."""
import os

import numpy as np
import pytest

from pyipm_amd.problems import example_problem, make_qp, unit_test_x0

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EPS = float(np.finfo(np.float64).eps)


def _close(a, b, scale, tol=1e-13):
    assert abs(a - b) <= tol * max(scale, 1e-300), (a, b, scale)


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10])
def test_merit_info_on_every_iterate_of_the_reference_traces(k):
    """Every Newton system of the reference's solve of example problem k: stage the point, take the step, and compare each
    entry of merit_info with the same quantity formed on the host in NumPy from the SAME device direction: ||c||_1, df.dx,
    sum ds/(s+eps), sum log s, the four KKT norms, s'lda, min s lda, |dx|, |ds| -- to 1e-13 relative to the sum of the
    magnitudes that enter each (the device adds in another order, nothing else differs).  Then the composite quantities
    the loop uses -- phi(x, s) - f(x), dphi, the nu threshold -- against the host IPM's own phi / dphi."""
    from pyipm_amd.ipm import IPM
    from pyipm_amd.newton import NewtonCore
    d = np.load(os.path.join(GOLD, "trace_p%02d.npz" % k))
    prob = example_problem(k)
    n, me, mi = prob["nvar"], prob["neq"], prob["nineq"]
    host = IPM(x0=unit_test_x0()[k], backend=object(), verbosity=-1,
               **{key: prob[key] for key in ("f", "df", "d2f", "ce", "dce", "d2ce", "ci", "dci", "d2ci") if prob.get(key) is not None})
    host.nvar, host.neq, host.nineq = n, me, mi
    core = NewtonCore(n, me, mi, device=0)
    for it in range(int(d["n_iter"])):
        x, s, lda, mu = d["it_x"][it], d["it_s"][it], d["it_lda"][it], float(d["it_mu"][it])
        d2L = np.array(prob["d2f"](x), dtype=np.float64)
        Je = ce = Ji = ci = None
        if me:
            d2L = d2L - prob["d2ce"](x, lda); Je, ce = prob["dce"](x), np.asarray(prob["ce"](x), dtype=float).reshape(me)
        if mi:
            d2L = d2L - prob["d2ci"](x, lda); Ji, ci = prob["dci"](x), np.asarray(prob["ci"](x), dtype=float).reshape(mi)
        df = np.asarray(prob["df"](x), dtype=float).reshape(n)
        core.stage_blocks(d2L, Je, Ji)
        core.stage_vectors(df, ce, ci, s if mi else None, lda if (me or mi) else None, mu=mu)
        delta = float(d["it_delta_out"][it]) if not np.array_equal(d["it_Hc"][it], d["it_H"][it]) else 0.0
        dz, _ = core.step(delta, 0.0)
        q = core.merit_info()
        g = core.residual().cpu().numpy()
        dz = dz.cpu().numpy()
        dx, ds = dz[:n], dz[n:n + mi]
        if me:
            _close(q["ce_l1"], np.abs(ce).sum(), np.abs(ce).sum())
            _close(q["kkt_ce"], np.linalg.norm(ce), np.linalg.norm(ce))
        if mi:
            r = ci - s
            _close(q["cis_l1"], np.abs(r).sum(), np.abs(r).sum())
            _close(q["kkt_ci"], np.linalg.norm(r), np.linalg.norm(r))
            _close(q["ds_over_s"], np.sum(ds / (s + EPS)), np.abs(ds / (s + EPS)).sum())
            _close(q["sum_log_s"], np.sum(np.log(s)), np.abs(np.log(s)).sum())
            _close(q["comp_sum"], np.sum(s * lda[me:]), np.abs(s * lda[me:]).sum())
            assert q["comp_min"] == np.min(s * lda[me:])
            _close(q["ds_norm"], np.linalg.norm(ds), np.linalg.norm(ds))
            _close(q["kkt_s"], np.linalg.norm(g[n:n + mi] * s), np.linalg.norm(g[n:n + mi] * s))
        _close(q["df_dx"], float(df @ dx), np.abs(df * dx).sum())
        _close(q["dx_norm"], np.linalg.norm(dx), np.linalg.norm(dx))
        _close(q["kkt_x"], np.linalg.norm(g[:n]), np.linalg.norm(g[:n]))
        # the KKT report is the reference's (pyipm.py:958-991): fixtures hold g = -grad per iterate
        _close(q["kkt_x"], np.linalg.norm(d["it_g"][it][:n]), max(np.linalg.norm(d["it_g"][it][:n]), 1.0), tol=1e-12)
        # composite quantities against the host loop's merit function at this iterate
        host.nu_host, host.mu_host_dev = 10.0, mu
        con = (q["ce_l1"] if me else 0.0) + (q["cis_l1"] if mi else 0.0)
        phi_minus_f = host.nu_host * con - (mu * q["sum_log_s"] if mi else 0.0)
        want = host.phi(x, s) - float(prob["f"](x))
        _close(phi_minus_f, want, abs(host.nu_host * con) + (abs(mu) * np.abs(np.log(s)).sum() if mi else 0.0) + abs(float(prob["f"](x))), tol=4e-13)
        dphi = q["df_dx"] - host.nu_host * con - (mu * q["ds_over_s"] if mi else 0.0)
        _close(dphi, host.dphi(x, s, dz[:n + mi]), np.abs(df * dx).sum() + host.nu_host * con + (mu * np.abs(ds / (s + EPS)).sum() if mi else 0.0))
    core.close()


def _qp_numpy_ray(qp, df, ce, ci, s, dz, nu, mu, alphas):
    """phi(x + a dx, s + a ds) - phi(x, s) in extended precision (the yardstick) from exactly what the device was given: the
    staged df, ce, ci, s (float64), its own direction, the blocks.  |c + a dc| - |c| is taken as sign(c) a dc wherever the
    sign does not change: evaluating the two magnitudes and subtracting would limit the yardstick itself to eps_longdouble |c|
    -- more than the whole change for the a ~ 1e-12 of a search's last candidates.
    Returns the values and, per candidate, the sum of the MAGNITUDES that enter it -- of the outer terms and of the inner
    products behind them (df . dx, dx'Q dx, A dx, G dx cancel internally: their rounding is relative to sum |a_i b_i|)."""
    n, me, mi = qp["n"], qp["me"], qp["mi"]
    L = np.longdouble
    Q = qp["Q"].astype(L)
    dx, ds = dz[:n].astype(L), dz[n:n + mi].astype(L)
    df, s = df.astype(L), (s.astype(L) if mi else None)

    def abs_change(c0, dc, a):
        c1 = c0 + a * dc
        same = ((c0 > 0) & (c1 >= 0)) | ((c0 < 0) & (c1 <= 0))
        return np.where(same, np.sign(c0) * a * dc, np.abs(c1) - np.abs(c0))

    g1, g2 = df @ dx, dx @ (Q @ dx)
    m1, m2 = np.abs(df * dx).sum(), np.abs(dx) @ (np.abs(Q) @ np.abs(dx))
    if me:
        A = qp["A"].astype(L)
        ce0, dce, mce = ce.astype(L), A @ dx, (np.abs(A) @ np.abs(dx)).sum()
    if mi:
        G = qp["G"].astype(L)
        r0, dr, mr = ci.astype(L) - s, G @ dx - ds, (np.abs(G) @ np.abs(dx)).sum() + np.abs(ds).sum()
    out, scale = [], []
    for a in alphas:
        a = L(a)
        v = a * g1 + a * a / 2 * g2
        sc = a * m1 + a * a / 2 * m2
        if me:
            v += nu * abs_change(ce0, dce, a).sum(); sc += nu * a * mce
        if mi:
            lg = np.log1p(a * ds / s)
            v += nu * abs_change(r0, dr, a).sum() - mu * lg.sum(); sc += nu * a * mr + mu * np.abs(lg).sum()
        out.append(float(v)); scale.append(float(sc))
    return np.array(out), np.array(scale)


@pytest.mark.parametrize("name", ["n40_me10_mi24_s11", "n64_me0_mi48_s12", "n48_me16_mi0_s13"])
def test_merit_ray_on_every_iterate_of_the_reference_qp_solves(name):
    """tests/golden/qptrace_*.npz (the UNMODIFIED reference solving QPs): at every iterate, the device ray -- ONE launch for
    the whole geometric sequence of backtracking candidates a0 tau^k -- against the closed form evaluated in extended
    precision from the problem data: <= 1e-13 of the summed magnitudes of the differences (the 'difference form': error
    relative to the CHANGE of phi, which is what an Armijo test near convergence needs; the magnitudes include those of the
    inner products behind every term, which cancel internally), and the same accept / reject
    decision of pyipm.py:1454-1459, 1534-1548 for every candidate as the host loop's two evaluations of phi give wherever
    that comparison is not inside the rounding of phi itself."""
    from pyipm_amd.ipm import IPM
    from pyipm_amd.newton import NewtonCore
    from pyipm_amd.problems import qp_callables
    d = np.load(os.path.join(GOLD, "qptrace_%s.npz" % name))
    n, me, mi, seed = int(d["n"]), int(d["me"]), int(d["mi"]), int(d["seed"])
    qp = make_qp(n, me, mi, seed)
    p = qp_callables(qp)
    host = IPM(x0=np.zeros(n), backend=object(), verbosity=-1, **{k: p[k] for k in ("f", "df", "d2f", "ce", "dce", "d2ce", "ci", "dci", "d2ci")
                                                                   if p.get(k) is not None})
    host.nvar, host.neq, host.nineq = n, me, mi
    core = NewtonCore(n, me, mi, device=0)
    core.stage_blocks(qp["Q"], qp["Je"] if me else None, qp["Ji"] if mi else None)
    tau, eta, nu = 0.995, 1e-4, 10.0
    checked = decided = 0
    for it in range(int(d["n_iter"])):
        x, s, lda, mu = d["it_x"][it], d["it_s"][it], d["it_lda"][it], float(d["it_mu_host"][it])
        df = qp["Q"] @ x + qp["c"]
        ce = (qp["A"] @ x - qp["b"]) if me else None
        ci = (qp["G"] @ x - qp["h"]) if mi else None
        core.stage_vectors(df, ce, ci, s if mi else None, lda if (me or mi) else None, mu=mu)
        dz, _ = core.step(float(d["it_delta_out"][it]), 0.0)
        a0 = core.step_lengths(tau)[0] if mi else 1.0
        alphas = [a0]
        for _ in range(199):
            alphas.append(alphas[-1] * tau)
        alphas += [1e-3 * a0, 1e-6 * a0, 1e-9 * a0, 1e-12 * a0]
        got = np.array(core.merit_ray(alphas, nu, mu))
        dzh = dz.cpu().numpy()
        want, scale = _qp_numpy_ray(qp, df, ce, ci, s, dzh, nu, mu, alphas)
        assert np.all(np.abs(got - want) <= 1e-13 * scale + 1e-300), (it, np.max(np.abs(got - want) / scale))
        # ... and relative to the CHANGE itself for the small steps of a search's end (no cancellation left in any term)
        tiny = np.abs(want) > 0
        assert np.all(np.abs(got - want)[-4:] <= 1e-10 * np.abs(want)[-4:]) or not tiny[-4:].all()
        checked += len(alphas)
        # Armijo decisions: host loop = two evaluations of phi (pyipm.py:1454-1459)
        q = core.merit_info()
        host.nu_host, host.mu_host_dev = nu, mu
        phi0 = host.phi(x, s if mi else np.zeros(0))
        con = (q["ce_l1"] if me else 0.0) + (q["cis_l1"] if mi else 0.0)
        dphi0 = q["df_dx"] - nu * con - (mu * q["ds_over_s"] if mi else 0.0)
        _close(dphi0, host.dphi(x, s if mi else np.zeros(0), dzh[:n + mi]),
               np.abs(df * dzh[:n]).sum() + nu * con + (mu * np.abs(dzh[n:n + mi] / (s + EPS)).sum() if mi else 0.0))
        for a, dk, sc in zip(alphas, got, scale):
            arm = phi0 + a * eta * dphi0
            host_val = host.phi(x + a * dzh[:n], (s + a * dzh[n:n + mi]) if mi else np.zeros(0))
            margin = abs(host_val - arm)
            if margin > 64 * EPS * max(abs(phi0), 1.0):          # (inside the rounding of phi the two forms may differ)
                assert ((phi0 + dk) - arm > 0.0) == (host_val - arm > 0.0), (it, a)
                decided += 1
    assert checked >= 204 * int(d["n_iter"]) and decided > checked // 2
    core.close()


def test_dots_and_quad_override():
    import torch
    from pyipm_amd.newton import NewtonCore
    rng = np.random.default_rng(2)
    n, me, mi = 1500, 300, 700
    qp = make_qp(n, me, mi, 8)
    core = NewtonCore(n, me, mi, device=0)
    core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
    core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
    a, b, c = (torch.from_numpy(rng.standard_normal(k)).cuda() for k in (n, n, 37))
    out = core.dots([(a, b), (c, c), (a, a)])
    for got, want, sc in zip(out, (a @ b, c @ c, a @ a), ((a * b).abs().sum(), c @ c, a @ a)):
        assert abs(got - float(want)) <= 1e-14 * float(sc)
    dz, _ = core.step(0.0, 0.0)
    alphas = [0.7, 0.3, 1e-4]
    ref = core.merit_ray(alphas, 10.0, 0.2)
    # an explicitly passed direction / quadratic term gives the same values (the L-BFGS mode's path: provider-only handle)
    dzc = dz.clone()
    qd = core.block_products(dzc[:n].contiguous())[0]
    quad = core.dots([(dzc[:n].contiguous(), qd)])[0]
    assert core.merit_ray(alphas, 10.0, 0.2, dz=dzc) == ref
    got = core.merit_ray(alphas, 10.0, 0.2, dz=dzc.clone(), quad=quad)
    assert np.allclose(got, ref, rtol=1e-14, atol=0)
    prov = NewtonCore(n, me, mi, device=0, provider_only=True)
    prov.stage_blocks(None, qp["Je"], qp["Ji"])
    prov.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
    assert np.allclose(prov.merit_ray(alphas, 10.0, 0.2, dz=dzc, quad=quad), ref, rtol=1e-14, atol=0)
    q1, q2 = core.merit_info(dz=dzc), prov.merit_info(dz=dzc)
    for key in ("ce_l1", "cis_l1", "df_dx", "ds_over_s", "sum_log_s", "comp_sum", "comp_min", "dx_norm", "ds_norm"):
        assert q1[key] == q2[key]
    assert np.isnan(q2["kkt_x"]) and np.isnan(prov.merit_info()["df_dx"])        # no residual formed / no direction on that handle
    core.close(); prov.close()


def test_search_phase_runs_on_the_library():
    """QPDeviceIPM's search phase: every reduction is a library launch -- counted through the timings the loop keeps (one
    merit_ray launch per search unless a search needs more than 64 candidates) -- and the solve is the one the torch-side
    merit function of rounds 1-3 produced (same iterates as the reference: test_gpu_qp.py retraces the fixtures)."""
    from pyipm_amd.qp import QPDeviceIPM
    qp = make_qp(300, 80, 160, 3)
    dev = QPDeviceIPM(qp["Q"], qp["c"], A=qp["A"], b=qp["b"], G=qp["G"], h=qp["h"], verbosity=-1, Ktol=1e-8, niter=30, miter=30)
    x, s, lda, f, kkt = dev.solve()
    assert dev.signal == 1 and max(kkt) <= 1e-8
    assert dev.timings["n_ray"] >= dev.iter_count and dev.timings["n_ray"] <= 3 * dev.iter_count
    r = qp["Q"] @ x.cpu().numpy() + qp["c"] - qp["A"].T @ lda.cpu().numpy()[:80] - qp["G"].T @ lda.cpu().numpy()[80:]
    assert np.linalg.norm(r) <= 1e-7
