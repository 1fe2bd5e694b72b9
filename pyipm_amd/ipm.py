"""Host-side ``IPM`` — the reference's user surface over the MI355X Newton-step core.

Mirrors ``/root/reference/pyipm.py``: constructor keywords (``:311-314``),
``solve(x0, s0, lda0, force_recompile) -> (x, s, lda, fval, kkt)`` (``:1567, 1863``),
``signal`` codes (``:1656, 1665, 1680, 1761, 1781, 1796, 1502, 1548``) and the
verbosity transcript (``README.md:101-122``).  The outer/inner barrier loop and the
line search stay host Python (restated here from the algorithm the reference
documents: ``:1597-1628`` init, ``:1408-1436`` fraction-to-boundary, ``:1438-1565``
backtracking search, ``:1727-1735`` merit parameter, ``:1804-1814`` barrier update,
``:958-991`` KKT report); the per-iteration Newton step ``:1717-1725`` is delegated
to a *Newton backend* — the HIP core (``pyipm_amd.newton``).  There is no CPU
implementation of that backend in the product: without the HIP library and a GPU
the constructor of the default backend raises.

Inputs.  Aesara is not required: ``f, df, d2f, ce, dce, d2ce, ci, dci, d2ci`` are
plain callables over NumPy arrays with the signatures of the reference's
"precompiled" functions (``:85-146, 512-562``):
``f(x)``, ``df(x)``, ``d2f(x)``, ``ce(x)``, ``dce(x)->(n,me)``, ``d2ce(x,lda)``,
``ci(x)``, ``dci(x)->(n,mi)``, ``d2ci(x,lda)``.  All derivatives must be supplied
(no autodiff).

``lbfgs=m`` (pyipm.py:195-203) switches the direction to the limited-memory quasi-Newton form
(``:1007-1371``): no second derivatives are needed, the host keeps the displacement storage
(``lbfgs_init`` / ``lbfgs_update``, O(n m)) and the direction itself — a Gram matrix over the n
variables, an (me+mi)-order symmetric factorisation and the Woodbury correction — runs on the device
behind ``pyipm_lbfgs_direction`` (``HipLbfgsBackend``).
"""
from __future__ import annotations

import numpy as np


def pinv_apply(J, g):
    """pinv(J) @ g -- the reference's first multiplier estimate (pyipm.py:726-730) -- without the SVD where J allows it: for a
    Jacobian of full rank and moderate condition the normal equations (Cholesky + one refinement step) give the same vector at
    a fraction of the cost (an SVD of a 1500 x 3600 Jacobian was half of a whole LP solve); rank-deficient or badly conditioned
    Jacobians (Cholesky fails or its diagonal spans more than three decades) take numpy.linalg.pinv as before."""
    import scipy.linalg
    J = np.asarray(J, dtype=np.float64)
    n, m = J.shape
    wide = m >= n
    G = J @ J.T if wide else J.T @ J
    try:
        cf = scipy.linalg.cho_factor(G, lower=True, check_finite=False)
        d = np.abs(np.diag(cf[0]))
        if d.size and np.isfinite(d).all() and d.min() > 1.0e-3 * d.max():
            def apply(r):
                if wide:
                    return J.T @ scipy.linalg.cho_solve(cf, r, check_finite=False)
                return scipy.linalg.cho_solve(cf, J.T @ r, check_finite=False)
            lam = apply(g)
            lam = lam + apply(g - J @ lam)
            # the spread of an unpivoted Cholesky diagonal does not bound cond(J J'): accept the vector only if it satisfies
            # the normal equations -- J lam = g for a wide J, J'(g - J lam) = 0 for a tall one -- to working precision
            # (ADVICE r4); anything else takes the SVD, whose rcond cutoff is what the reference's pinv applies
            r = g - J @ lam
            if not wide:
                r = J.T @ r
            scale = np.linalg.norm(g) if wide else np.linalg.norm(J) * np.linalg.norm(g)      # (Frobenius: no SVD)
            if np.isfinite(lam).all() and np.linalg.norm(r) <= 1.0e-11 * max(scale, np.finfo(float).tiny):
                return lam
    except (np.linalg.LinAlgError, ValueError):
        pass
    return np.linalg.pinv(J) @ g


class HipNewtonBackend(object):
    """Newton backend on the HIP core: the counterpart of ``reghess`` + ``sym_solve_cmp``
    (pyipm.py:1373-1406, 1717-1725).

    * Inertia comes from the block pivots of the factorisation instead of ``eigvalsh`` (Sylvester).
    * The reference solves with LU + partial pivoting over the WHOLE matrix (pyipm.py:18-20); the device pivots
      inside 64x64 tiles.  A pivot that cannot be avoided inside its tile and has cancelled to nothing (LPs and
      other problems with zero Hessian rows: the x-x tile is singular although the KKT matrix is not) becomes a
      *static pivot* (``n_zero``): the factor is that of a matrix perturbed by sqrt(eps)|Hc| in those diagonal
      entries, and the direction of the UNperturbed system is recovered by refinement against the KKT blocks
      (``solve(refine=-1)``).  Converged => the reference's own answer with its own "no shift" decision.  Not
      converged => Hc is singular to working precision, which is what the reference's ``rcond <= eps`` test
      detects (pyipm.py:1379-1381): the same delta / delta_c branch is taken.
    * Without static pivots the "rcond <= eps" trigger is "d_min/d_max <= eps" on the block pivots."""

    berr_tol = 1e-11                       # backward error a refined direction must meet to count as converged
    berr_fallback = float(np.sqrt(np.finfo(float).eps))   # ... and the bar of the LAST resort: when no shift within the budget
                                           # yields a converged direction, the best one seen is returned if it meets this (the
                                           # reference returns its LU direction whatever its accuracy, pyipm.py:1720-1721) and
                                           # counted in n_inexact; only beyond it the backend gives up

    def __init__(self, n, me, mi, device=None, nb=256, refine=0, max_shift_tries=60, device_step=False,
                 condensed=False):
        from .newton import NewtonCore
        self.core = NewtonCore(n, me, mi, device=device, nb=nb)
        self.condensed_requested = bool(condensed and mi)
        self.condensed_on = self.condensed_requested
        self.condensed_tol = 1e-9           # backward-error bar a condensed direction must meet
        self.n_condensed_fallback = 0
        self.condensed_fallback_reason = None
        self.n_calls = 0
        if self.condensed_on:               # factor the (n+me) condensed system
            self.core.set_option("condensed", 1)
        self.n, self.me, self.mi = n, me, mi
        self.refine = refine
        self.device_step = device_step      # closed-form step lengths on the device
        self.max_shift_tries = max_shift_tries
        self.n_factor = 0
        self.n_static = 0                   # directions recovered from a statically pivoted factor
        self.n_unconverged = 0              # refined solves that missed berr_tol and were sent to the shift branch
        self.n_inertia_retries = 0          # shifted factorisations whose inertia was still wrong (delta *= 10, pyipm.py:1399-1403)
        self.n_inexact = 0                  # directions returned on berr_fallback after the shift budget was spent
        self.rcond_log = []                 # (call, rcond estimate, pivot spread) of every estimate taken
        self.last_solve_info = None

    def shape(self):
        return (self.n, self.me, self.mi)

    def new_solve(self):
        """Called at the start of every IPM.solve(): a condensed-form fallback lasts for one solve, not for the
        lifetime of the backend."""
        if self.condensed_requested and not self.condensed_on:
            self.condensed_on = True
            self.core.set_option("condensed", 1)

    def _factor(self):
        """factor(); a NaN/Inf met on the way is reported like a singular matrix (the host regularises and retries,
        as reghess would) instead of aborting the solve."""
        from .newton import NewtonError
        self.n_factor += 1
        try:
            return self.core.factor()
        except NewtonError as e:
            if e.code != -4 or e.stats is None:
                raise
            return e.stats

    suspect_spread = 1e-10                 # pivot spread d_min/d_max below which the condition estimate is consulted

    def _singular(self, st, eps):
        """The reference's ``rcond <= eps`` test (pyipm.py:1379-1381, rcond = min|w|/max|w| over the eigenvalues).
        Block pivots of moderate spread cannot hide an eigenvalue ratio at the eps level; otherwise, and whenever
        static pivots were placed, the ratio is ESTIMATED from the factor and the blocks (``core.rcond``: inverse /
        power iterations).  With static pivots the factor is that of a perturbed matrix, whose smallest eigenvalue
        sits at the perturbation level exactly when the unperturbed matrix is singular.

        The estimate is taken with adaptive iteration counts (``core.rcond(-1, -1)``, include/pyipm_newton.h): late in an
        interior-point run the pivot spread stays below ``suspect_spread`` at EVERY iterate (Sigma = lda_i / s spans twenty
        decades by itself, the spread falls a thousandfold per iterate while rcond falls twentyfold: 2e-6 ... 3e-16 over the
        last seven iterates of the benchmark QP), so the estimate is part of every late step: 7.4 ms instead of 14.8 at
        N = 32768, and it leaves a right-hand side that was forward-substituted under the factorisation in place."""
        self.last_rcond = None
        if st["nonfinite"]:
            return True
        spread = st["d_min"] / st["d_max"] if st["d_max"] > 0 else 1.0
        if st["n_zero"] == 0 and spread > self.suspect_spread:
            return False
        # adaptive iteration counts (include/pyipm_newton.h): the threshold decision of the fixed 3 + 6 iterations at a
        # third of their cost when the matrix is far from singular -- every suspect iterate of a convex QP
        est = self.last_rcond = self.core.rcond(-1, -1) if st["n_zero"] == 0 else self.core.rcond()
        self.n_rcond += 1
        self.rcond_log.append((self.n_calls, est["rcond"], spread))
        if st["n_zero"] > 0 and est["w_min"] <= 100.0 * est["static_pivot"]:
            return True
        return est["rcond"] <= eps

    n_rcond = n_rcond_reused = 0
    rcond_log = []

    @staticmethod
    def _at_risk(st):
        """Block pivots whose accuracy the tile-local pivot search does not guarantee: static pivots, 2x2 pivots
        (indefinite tiles) or large entries of L.  Such a direction is checked against the blocks and refined."""
        return st["n_zero"] > 0 or st["n_2x2"] > 0 or st["growth"] > 64.0

    def _solve(self, st):
        """Substitution for the current factor; (dz tensor, converged)."""
        core = self.core
        if self._at_risk(st):
            dz = core.solve(flip=True, refine=-1)
            info = self.last_solve_info = core.solve_info()
            return dz, info["backward_error"] >= 0.0 and info["backward_error"] <= self.berr_tol
        self.last_solve_info = None
        return core.solve(flip=True, refine=self.refine), True

    def direction(self, d2L, Je, Ji, df, ce, ci, s, lda, mu, delta, mu_host, eta, beta, reg_coef, delta0, eps,
                  as_tensor=False, staged=False, g=None):
        """as_tensor: inputs may be device tensors (staged without copies) and dz stays on the device.
        staged: the caller has already staged the blocks (constant for a QP: once per solve) and this point's vectors and
        formed g = -grad in the handle (``core.residual()``, passed as ``g``) -- nothing is staged again here."""
        core, need = self.core, self.me + self.mi
        self.n_calls += 1
        self._dir_override = None          # set when the direction returned is NOT the handle's last solve (see step_lengths)
        if not staged:
            core.stage_blocks(d2L, Je, Ji)
            core.stage_vectors(df, ce, ci, s, lda, mu=mu, eps=eps)
            g = core.residual()
        elif g is None and self.condensed_on:
            g = core.residual()
        if self.condensed_on:
            # Condensed system first (2x fewer flops at the benchmark shape).  The block pivots are explicit
            # 64x64 inverses, so a dense ill-conditioned tile (Sigma spanning > ~1e8 late in a run) costs
            # accuracy ~ eps*sqrt(cond): accept only a direction whose backward error against the FULL blocks
            # is small and whose inertia is right; otherwise use the full system for the rest of this solve.
            core.assemble(0.0, 0.0)
            st = self._factor()
            ok = st["n_zero"] == 0 and st["n_neg"] == need and not st["nonfinite"]
            why = "inertia %d/%d, %d static pivots" % (st["n_neg"], need, st["n_zero"])
            if ok:
                gn = float(g.norm())
                for refine in (self.refine, max(self.refine, 1) + 3):      # second try: more refinement steps
                    dz = core.solve(flip=True, refine=refine)
                    raw = dz.clone()
                    raw[self.n + self.mi:] *= -1.0
                    rn = float((core.matvec(raw) - g).norm())
                    berr = rn / gn if gn > 0.0 else rn                     # zero residual: the zero direction is exact
                    ok = berr <= self.condensed_tol
                    why = "backward error %.1e after %d refinement steps" % (berr, max(refine, 1))
                    if ok:
                        break
            if ok:
                if not as_tensor:
                    dz = dz.cpu().numpy()
                return dz, 0.0 if delta == 0.0 else float(delta), st
            self.condensed_on = False
            self.n_condensed_fallback += 1
            self.condensed_fallback_reason = "call %d: %s" % (self.n_calls, why)
            core.set_option("condensed", 0)
        core.assemble(0.0, 0.0)
        st = self._factor()
        singular = self._singular(st, eps)
        dz = None
        best = None                        # (backward error, dz, delta, stats) of the best unconverged direction so far

        def remember(dz_, delta_, st_):
            nonlocal best
            info = self.last_solve_info
            be = info["backward_error"] if info else -1.0
            # the LEAST shifted direction that meets berr_fallback is kept (the reference does not shift such a system at
            # all); below that bar, the most accurate one
            if dz_ is not None and be >= 0.0 and np.isfinite(be) and \
                    (best is None or (best[0] > self.berr_fallback and be < best[0])):
                best = (be, dz_.clone() if hasattr(dz_, "clone") else dz_, float(delta_), st_)

        if not singular and st["n_neg"] == need:
            dz, converged = self._solve(st)
            if not converged:
                # a statically pivoted / 2x2 / high-growth factor whose refinement against the blocks stalls above
                # berr_tol solves a nearby system, not this one: what the reference's LU would flag as rcond <= eps.
                # Take the branch reghess takes then (pyipm.py:1379-1403) instead of returning the direction silently.
                self.n_unconverged += 1
                remember(dz, 0.0, st)
                dz, singular = None, True
            elif st["n_zero"] > 0:
                self.n_static += 1         # reference: LU over the whole matrix, no shift (pyipm.py:1381 not taken)
        if dz is None:
            delta_c = reg_coef * eta * (mu_host ** beta) if (singular and self.me) else 0.0
            delta = delta0 if delta == 0.0 else max(delta / 2.0, delta0)
            tries = 0
            while True:
                core.assemble(delta, delta_c)
                st = self._factor()
                if st["n_neg"] == need and not st["nonfinite"]:
                    break
                tries += 1
                self.n_inertia_retries += 1
                if tries > self.max_shift_tries:
                    raise RuntimeError("inertia not corrected after %d diagonal shifts" % tries)
                delta *= 10.0
            dz, converged = self._solve(st)
            stalled, prev_be = 0, None
            solved = True                  # the loop head looks at a solve that really happened (not at an inertia retry)
            while not converged:
                # still no direction that satisfies the blocks: the shift has not made the factor trustworthy yet.
                # Larger shift (the reference's delta *= 10 loop, :1399-1403); when the budget is spent, the best direction
                # seen if it is at least berr_fallback-accurate (shifted or not), else give up loudly.
                if solved:
                    self.n_unconverged += 1
                    remember(dz, delta, st)
                    # ... or as soon as larger shifts have stopped helping: two SOLVES in a row whose backward error did not
                    # fall tenfold with the tenfold shift, and a direction at berr_fallback in hand.  (Exactly dependent
                    # equality constraints: no shift of the x block cures the multiplier block; waiting for delta ~ 1 to
                    # dominate the system cost nine factorisations per iterate and returned a gradient-like step -- round 4.)
                    # A factorisation whose inertia was wrong produced no direction: it takes no part in this bookkeeping
                    # (two inertia retries used to count as "larger shifts stopped helping").
                    be = self.last_solve_info["backward_error"] if self.last_solve_info else -1.0
                    stalled = stalled + 1 if (prev_be is not None and be >= 0.0 and be >= 0.1 * prev_be) else 0
                    prev_be = be if be >= 0.0 else prev_be
                    if stalled >= 2 and best is not None and best[0] <= self.berr_fallback:
                        self.n_inexact += 1
                        dz, delta, st = best[1], best[2], best[3]
                        self._dir_override = dz
                        break
                tries += 1
                if tries > self.max_shift_tries:
                    if best is not None and best[0] <= self.berr_fallback:
                        self.n_inexact += 1
                        dz, delta, st = best[1], best[2], best[3]
                        self._dir_override = dz
                        break
                    raise RuntimeError("refined solve did not reach backward error %.1e after %d diagonal shifts (last: %s)"
                                       % (self.berr_tol, tries, self.last_solve_info))
                delta *= 10.0
                core.assemble(delta, delta_c)
                st = self._factor()
                if st["n_neg"] != need or st["nonfinite"]:
                    self.n_inertia_retries += 1      # (an inertia retry, not an unconverged solve: counted on its own)
                    solved = False
                    continue
                dz, converged = self._solve(st)
                solved = True
        if not as_tensor:
            dz = dz.cpu().numpy()
        return dz, float(delta), st

    _dir_override = None

    def step_lengths(self, tau):
        """(alpha_smax, alpha_lmax) for the direction just returned, or None to let the host search.  When direction() gave up
        on shifting and returned an EARLIER, less shifted solve (n_inexact), the handle's last direction is another one: the
        lengths are then taken for the direction actually returned (they used to come from the most shifted
        solve, and s + a ds could leave the positive orthant)."""
        if not (self.device_step and self.mi):
            return None
        return self.core.step_lengths(tau, dz=self._dir_override)


class HipLbfgsBackend(object):
    """L-BFGS direction on the HIP library: the counterpart of ``lbfgs_dir_func`` (pyipm.py:872-875,
    1184-1246).  The "rcond <= eps" trigger on the equality block (:1108-1113, an ``eigh`` in the
    reference) is taken from the block pivots of the factorisation, as in ``HipNewtonBackend``."""

    def __init__(self, n, me, mi, memory, device=None, nb=256, linear_constraints=False):
        from .lbfgs import LbfgsCore
        self.core = LbfgsCore(n, me, mi, int(memory) + 1, device=device, nb=nb)   # storage grows to memory+1 (:1300)
        self.n, self.me, self.mi = n, me, mi
        self.linear_constraints = bool(linear_constraints)   # dce/dci do not depend on x: stage once, J'J is reused
        self.n_calls = self.n_staged = 0

    def shape(self):
        return (self.n, self.me, self.mi)

    def lbfgs_direction(self, Je, Ji, s, lda, g, zeta, S, Y, SS, L, D, reg, eps):
        self.n_calls += 1
        if (self.me or self.mi) and not (self.linear_constraints and self.n_staged):
            self.core.stage_jacobian(Je, Ji)
            self.n_staged += 1
        dz, self.last_stats = self.core.direction(g, s, lda, zeta, S, Y, SS, L, D, reg=reg, eps=eps, flip=False)
        if S.shape[1] and self.last_stats["small_pivot_min"] == 0.0:
            # the reference's sym_solve raises on an exactly singular system (scipy.linalg.solve, pyipm.py:18-20)
            raise np.linalg.LinAlgError("L-BFGS: the 2m x 2m system is singular (zero pivot)")
        return dz.cpu().numpy()


class IPM(object):
    """Line-search primal-dual interior-point solver; see module docstring."""

    def __init__(self, x0=None, x_dev=None, f=None, df=None, d2f=None, ce=None, dce=None, d2ce=None, ci=None,
                 dci=None, d2ci=None, lda0=None, lambda_dev=None, s0=None, mu=0.2, nu=10.0, rho=0.1, tau=0.995,
                 eta=1.0E-4, beta=0.4, miter=20, niter=10, Xtol=None, Ktol=1.0E-4, Ftol=None, lbfgs=False,
                 lbfgs_zeta=None, float_dtype=np.float64, verbosity=1, backend=None, device=None, nb=256, refine=0,
                 device_step=False, condensed=False, linear_constraints=False):
        self.x0, self.s0, self.lda0 = x0, s0, lda0
        self.x_dev, self.lambda_dev = x_dev, lambda_dev        # accepted for signature parity; unused
        self.f, self.df, self.d2f = f, df, d2f
        self.ce, self.dce, self.d2ce = ce, dce, d2ce
        self.ci, self.dci, self.d2ci = ci, dci, d2ci
        self.nvar = self.neq = self.nineq = None
        self.float_dtype = float_dtype
        self.eps = float(np.finfo(float_dtype).eps)
        self.mu, self.nu, self.rho, self.tau, self.eta, self.beta = mu, nu, rho, tau, eta, beta
        self.miter, self.niter = miter, niter
        self.Xtol = Xtol if Xtol else self.eps
        self.Ktol, self.Ftol = Ktol, Ftol
        self.reg_coef = float(np.sqrt(self.eps))
        self.delta0 = self.reg_coef
        self.lbfgs = lbfgs
        self.lbfgs_zeta = 1.0 if (lbfgs and lbfgs_zeta is None) else lbfgs_zeta         # pyipm.py:355-359
        self.lbfgs_fail_max = lbfgs                                                     # :360
        self.verbosity = verbosity
        self.backend = backend
        self._own_backend = backend is None
        self._backend_opts = dict(device=device, nb=nb, refine=refine, device_step=device_step, condensed=condensed)
        self.linear_constraints = linear_constraints       # L-BFGS mode: Jacobians staged once, J'J reused
        self.compiled = False
        self.signal = 0

    # ------------------------------------------------------------------ setup
    def validate(self):
        assert self.f is not None
        assert (self.ce is not None) or (self.dce is None and self.d2ce is None)
        assert (self.ci is not None) or (self.dci is None and self.d2ci is None)
        assert self.mu > 0.0 and self.nu > 0.0
        assert 0.0 < self.eta < 1.0 and 0.0 < self.rho < 1.0 and 0.0 < self.tau < 1.0
        assert self.beta < 1.0
        assert isinstance(self.miter, int) and self.miter >= 0
        assert isinstance(self.niter, int) and self.niter >= 0
        assert self.Xtol >= self.eps and self.Ktol >= self.eps
        assert self.Ftol is None or self.Ftol >= 0.0
        assert self.lbfgs is False or self.lbfgs >= 0                                  # pyipm.py:405-408
        if self.lbfgs:
            assert isinstance(self.lbfgs, int) and not isinstance(self.lbfgs, bool)
            if self.lbfgs > 31:
                raise NotImplementedError("the device direction keeps the 2m x 2m system in one 64 x 64 tile: lbfgs <= 31")
        assert self.lbfgs_zeta is None or self.lbfgs_zeta > 0.0
        if self.float_dtype != np.float64:
            raise NotImplementedError("the Newton-step core computes in fp64 only")
        second = () if self.lbfgs else ("d2f",)
        for name in ("df",) + second:
            if getattr(self, name) is None:
                raise ValueError("%s must be supplied as a callable (no autodiff without Aesara)" % name)
        if self.ce is not None and (self.dce is None or (self.d2ce is None and not self.lbfgs)):
            raise ValueError("dce and d2ce must be supplied with ce")
        if self.ci is not None and (self.dci is None or (self.d2ci is None and not self.lbfgs)):
            raise ValueError("dci and d2ci must be supplied with ci")

    def compile(self, nvar=None, neq=None, nineq=None):
        """Infer constraint counts (pyipm.py:443-467) and bind the Newton backend."""
        if nvar is not None:
            self.nvar = nvar
        x0 = np.asarray(self.x0, dtype=np.float64)
        self.neq = int(np.size(self.ce(x0))) if (self.ce is not None and neq is None) else int(neq or 0)
        self.nineq = int(np.size(self.ci(x0))) if (self.ci is not None and nineq is None) else int(nineq or 0)
        if self.backend is not None and self._own_backend and hasattr(self.backend, "shape") and \
                self.backend.shape() != (self.nvar, self.neq, self.nineq):
            self.backend = None            # force_recompile with another problem shape: the handle is per shape
        if self.backend is None:
            self._own_backend = True
            if self.lbfgs:
                self.backend = HipLbfgsBackend(self.nvar, self.neq, self.nineq, self.lbfgs,
                                               device=self._backend_opts["device"], nb=self._backend_opts["nb"],
                                               linear_constraints=self.linear_constraints)
            else:
                self.backend = HipNewtonBackend(self.nvar, self.neq, self.nineq, **self._backend_opts)
        self.compiled = True

    # ------------------------------------------------------------------ model pieces (host, O(n m))
    def _con(self, x, s):
        parts = []
        if self.neq:
            parts.append(np.asarray(self.ce(x), dtype=np.float64).reshape(self.neq))
        if self.nineq:
            parts.append(np.asarray(self.ci(x), dtype=np.float64).reshape(self.nineq) - s)
        return np.concatenate(parts) if parts else np.zeros(0)

    def _jac_x(self, x):
        cols = []
        if self.neq:
            cols.append(np.asarray(self.dce(x), dtype=np.float64).reshape(self.nvar, self.neq))
        if self.nineq:
            cols.append(np.asarray(self.dci(x), dtype=np.float64).reshape(self.nvar, self.nineq))
        return np.concatenate(cols, axis=1)

    def _jaco(self, x):
        """(n+mi) x (me+mi) composite Jacobian [[Je, Ji],[0, -I]] (pyipm.py:582-607)."""
        top = self._jac_x(x)
        if not self.nineq:
            return top
        bottom = np.concatenate([np.zeros((self.nineq, self.neq)), -np.eye(self.nineq)], axis=1)
        return np.concatenate([top, bottom], axis=0)

    def grad(self, x, s, lda):
        """KKT residual on the host (same formula as K2; used by the KKT report and convergence tests)."""
        n, me, mi = self.nvar, self.neq, self.nineq
        gx = np.array(self.df(x), dtype=np.float64).reshape(n)
        out = [None]
        if me:
            gx = gx - np.asarray(self.dce(x), dtype=np.float64).reshape(n, me) @ lda[:me]
        if mi:
            gx = gx - np.asarray(self.dci(x), dtype=np.float64).reshape(n, mi) @ lda[me:]
            out.append(lda[me:] - self.mu_host_dev / (s + self.eps))
        out[0] = gx
        if me:
            out.append(np.asarray(self.ce(x), dtype=np.float64).reshape(me))
        if mi:
            out.append(np.asarray(self.ci(x), dtype=np.float64).reshape(mi) - s)
        return np.concatenate(out)

    def KKT(self, x, s, lda):
        """First-order conditions split in four blocks, slack block scaled by s (pyipm.py:958-991)."""
        n, me, mi = self.nvar, self.neq, self.nineq
        k = self.grad(x, s, lda)
        zero = np.float64(0.0)
        k1 = k[:n]
        k2 = k[n:n + mi] * s if mi else zero
        k3 = k[n + mi:n + mi + me] if me else zero
        k4 = k[n + mi + me:] if mi else zero
        return k1, k2, k3, k4

    def phi(self, x, s):
        v = float(self.f(x))
        if self.neq:
            v += self.nu_host * np.sum(np.abs(self.ce(x)))
        if self.nineq:
            v += self.nu_host * np.sum(np.abs(np.asarray(self.ci(x)).reshape(self.nineq) - s))
            v -= self.mu_host_dev * np.sum(np.log(s))
        return v

    def dphi(self, x, s, dz):
        n = self.nvar
        v = float(np.dot(np.asarray(self.df(x)).reshape(n), dz[:n]))
        if self.neq:
            v -= self.nu_host * np.sum(np.abs(self.ce(x)))
        if self.nineq:
            v -= self.nu_host * np.sum(np.abs(np.asarray(self.ci(x)).reshape(self.nineq) - s))
            v -= float(np.dot(self.mu_host_dev / (s + self.eps), dz[n:]))
        return v

    # ------------------------------------------------------------------ line-search pieces
    def step(self, x, dx):
        """Fraction-to-the-boundary step by golden-section search on [0,1] (pyipm.py:1408-1436)."""
        gold = (np.sqrt(5.0) + 1.0) / 2.0
        floor = (1.0 - self.tau) * x

        def feasible(alpha):
            return bool(np.all(x + alpha * dx >= floor))

        lo, hi = 0.0, 1.0
        if feasible(hi):
            return hi
        c = hi - (hi - lo) / gold
        d = lo + (hi - lo) / gold
        while abs(hi - lo) > gold * self.Xtol:
            if feasible(d):
                lo = d
            else:
                hi = d
            if c > lo:
                if feasible(c):
                    lo = c
                else:
                    hi = c
            c = hi - (hi - lo) / gold
            d = lo + (hi - lo) / gold
        return lo

    def _restoration(self, x0, c_new):
        """Feasibility-restoration direction of the second-order correction (pyipm.py:1466-1477,
        1518-1529): a square solve when the Jacobian happens to be square, otherwise (the usual
        case) the minimum-norm least-squares solution."""
        A = self._jaco(x0).T
        if A.shape[0] == A.shape[1]:
            try:
                import scipy.linalg
                return -scipy.linalg.solve(A, c_new.reshape(-1, 1)).reshape(-1)
            except Exception:
                pass
        return -np.linalg.lstsq(A, c_new, rcond=None)[0]

    def search(self, x0, s0, lda0, dz, alpha_smax, alpha_lmax):
        """Backtracking Armijo search on the merit function with an optional second-order
        feasibility correction (pyipm.py:1438-1565)."""
        n, me, mi = self.nvar, self.neq, self.nineq
        dx = dz[:n]
        ds = dz[n:n + mi] if mi else np.zeros(0)
        if me or mi:
            dl = dz[n + mi:]
        else:
            dl, alpha_lmax = 0.0, 0.0
        phi0 = self.phi(x0, s0)
        dphi0 = self.dphi(x0, s0, dz[:n + mi])
        armijo = lambda a: phi0 + a * self.eta * dphi0    # noqa: E731
        corrected, alpha_corr, dz_p = False, 1.0, None

        def trial(a):
            return self.phi(x0 + a * dx, s0 + a * ds) if mi else self.phi(x0 + a * dx, s0)

        if trial(alpha_smax) > armijo(alpha_smax):
            if me or mi:
                c_old = self._con(x0, s0)
                c_new = self._con(x0 + alpha_smax * dx, s0 + alpha_smax * ds if mi else s0)
                if np.sum(np.abs(c_new)) > np.sum(np.abs(c_old)):
                    dz_p = self._restoration(x0, c_new)
                    if mi:
                        xs = x0 + alpha_smax * dx + dz_p[:n]
                        ss = s0 + alpha_smax * ds + dz_p[n:]
                        if self.phi(xs, ss) <= armijo(alpha_smax):
                            alpha_corr = self.step(s0, alpha_smax * ds + dz_p[n:])
                            if (self.phi(x0 + alpha_corr * (alpha_smax * dx + dz_p[:n]),
                                         s0 + alpha_corr * (alpha_smax * ds + dz_p[n:])) <= armijo(alpha_smax)):
                                corrected = True
                    else:
                        if self.phi(x0 + alpha_smax * dx + dz_p[:n], s0) <= armijo(alpha_smax):
                            alpha_corr, corrected = 1.0, True
                    if corrected and self.verbosity > 2:
                        print('Second-order feasibility correction accepted')
            if not corrected:
                alpha_smax *= self.tau
                alpha_lmax *= self.tau
                while trial(alpha_smax) > armijo(alpha_smax):
                    size = np.linalg.norm(alpha_smax * dx)
                    if mi:
                        size = np.sqrt(size ** 2 + np.linalg.norm(alpha_lmax * ds) ** 2)
                    if size < self.eps:
                        if self.verbosity > 2:
                            print('Search direction is unreliable to machine precision.')
                        self.signal = -2
                        return x0, s0, lda0
                    alpha_smax *= self.tau
                    alpha_lmax *= self.tau
        if corrected:
            x = x0 + alpha_corr * (alpha_smax * dx + dz_p[:n])
            s = s0 + alpha_corr * (alpha_smax * ds + dz_p[n:]) if mi else np.copy(s0)
        else:
            x = x0 + alpha_smax * dx
            s = s0 + alpha_smax * ds if mi else np.copy(s0)
        lda = lda0 + alpha_lmax * dl if (me or mi) else np.copy(lda0)
        return x, s, lda

    # ------------------------------------------------------------------ the Newton step (hot path)
    def newton_direction(self, x, s, lda):
        """Counterpart of pyipm.py:1717-1725: host evaluates the derivative blocks, the backend
        assembles / regularises / factors / solves on the device and returns dz (sign-flipped)."""
        n, me, mi = self.nvar, self.neq, self.nineq
        d2L = np.array(self.d2f(x), dtype=np.float64).reshape(n, n)
        Je = Ji = ce = ci = None
        if me:
            d2L = d2L - np.asarray(self.d2ce(x, lda), dtype=np.float64).reshape(n, n)
            Je = np.asarray(self.dce(x), dtype=np.float64).reshape(n, me)
            ce = np.asarray(self.ce(x), dtype=np.float64).reshape(me)
        if mi:
            d2L = d2L - np.asarray(self.d2ci(x, lda), dtype=np.float64).reshape(n, n)
            Ji = np.asarray(self.dci(x), dtype=np.float64).reshape(n, mi)
            ci = np.asarray(self.ci(x), dtype=np.float64).reshape(mi)
        df = np.asarray(self.df(x), dtype=np.float64).reshape(n)
        dz, self.delta, self.last_stats = self.backend.direction(
            d2L, Je, Ji, df, ce, ci, s, lda, self.mu_host_dev, self.delta, self.mu_host, self.eta, self.beta,
            self.reg_coef, self.delta0, self.eps)
        return dz

    # ------------------------------------------------------------------ L-BFGS storage (host, O(n m))
    def lbfgs_init(self):
        """Empty displacement storage (pyipm.py:993-1005)."""
        n, z = self.nvar, np.zeros
        return float(self.lbfgs_zeta), z((n, 0)), z((n, 0)), z((0, 0)), z((0, 0)), z((0, 0)), 0

    def lbfgs_update(self, x_old, x_new, g_old, g_new, zeta, S, Y, SS, L, D, lbfgs_fail):
        """Append the newest (dx, dg) pair and its inner products; pairs of non-positive curvature are
        skipped and counted, too many skips in a row reset the storage (pyipm.py:1282-1371).  For constrained
        problems SS = S'S, L = strictly-lower(S'Y), zeta scales the Hessian; for unconstrained ones the same
        arrays hold Y'Y and the upper-triangular S'Y and zeta scales the inverse Hessian."""
        n, con = self.nvar, bool(self.neq or self.nineq)
        dx = x_new - x_old
        dg = g_old[:n] - g_new[:n]
        curv = float(np.dot(dg, dx))
        zeta_new = curv / ((np.dot(dx, dx) if con else np.dot(dg, dg)) + self.eps)
        root = np.sqrt(self.eps)
        if curv > root and zeta_new > root:
            zeta = zeta_new
            k = S.shape[1]
            if k > self.lbfgs:                           # the reference lets the storage reach lbfgs+1 pairs (:1300)
                S, Y = np.roll(S, -1, axis=1), np.roll(Y, -1, axis=1)
                SS, L, D = (np.roll(Mx, (-1, -1), axis=(0, 1)) for Mx in (SS, L, D))
                SS[-1, :] = SS[:, -1] = 0.0
                L[-1, :] = L[:, -1] = 0.0
                D[-1, :] = D[:, -1] = 0.0
            else:
                S, Y = np.pad(S, ((0, 0), (0, 1))), np.pad(Y, ((0, 0), (0, 1)))
                SS, L, D = (np.pad(Mx, ((0, 1), (0, 1))) for Mx in (SS, L, D))
            S[:, -1], Y[:, -1] = dx, dg
            inner = S.T @ dx if con else Y.T @ dg
            SS[:, -1] = SS[-1, :] = inner
            if con:
                L[-1, :] = dx @ Y
                L[-1, -1] = 0.0
            else:
                L[:, -1] = S.T @ dg
            D[-1, -1] = curv
            lbfgs_fail = 0
        else:
            lbfgs_fail += 1
        if lbfgs_fail > self.lbfgs_fail_max and S.shape[1] > 0:
            if self.verbosity > 2:
                print('Max failures reached, resetting storage arrays.')
            return self.lbfgs_init()
        return zeta, S, Y, SS, L, D, lbfgs_fail

    def lbfgs_dir(self, x, s, lda, g, zeta, S, Y, SS, L, D):
        """Counterpart of pyipm.py:1184-1246: the host evaluates the constraint Jacobians, the backend forms
        and solves the Woodbury system on the device and returns the RAW direction (the caller flips the
        multiplier rows, :1723-1725).  The square-Jacobian shortcut (:1188-1198) is not taken: it returns the
        same direction through inv(B) and, as written, cannot be compiled (duplicate input, :877-880)."""
        n, me, mi = self.nvar, self.neq, self.nineq
        Je = np.asarray(self.dce(x), dtype=np.float64).reshape(n, me) if me else None
        Ji = np.asarray(self.dci(x), dtype=np.float64).reshape(n, mi) if mi else None
        reg = self.reg_coef * self.eta * (self.mu_host_dev ** self.beta)              # :1113
        return self.backend.lbfgs_direction(Je, Ji, s, lda, g, zeta, S, Y, SS, L, D, reg, self.eps)

    # ------------------------------------------------------------------ driver
    def _kkt_small(self, kkt, tol):
        return all(np.linalg.norm(k) <= tol for k in kkt)

    def solve(self, x0=None, s0=None, lda0=None, force_recompile=False):
        if x0 is not None:
            self.x0 = x0
        if s0 is not None:
            self.s0 = s0
        if lda0 is not None:
            self.lda0 = lda0
        assert self.x0 is not None and np.size(self.x0) > 0
        self.x0 = np.asarray(self.x0, dtype=np.float64)
        assert self.x0.ndim == 1
        self.nvar = self.x0.size
        self.validate()
        if not self.compiled or force_recompile:
            self.compile()
        if hasattr(self.backend, "new_solve"):
            self.backend.new_solve()
        n, me, mi = self.nvar, self.neq, self.nineq

        # initial point (pyipm.py:1597-1625)
        x = self.x0
        if mi:
            s = (np.maximum(np.asarray(self.ci(x), dtype=np.float64).reshape(mi), self.Ktol)
                 if self.s0 is None else np.asarray(self.s0, dtype=np.float64))
            self.mu_host = self.mu
            self.mu_host_dev = self.mu            # value of the reference's shared mu_dev
        else:
            s = np.zeros(0)
            self.mu_host = self.Ktol
            self.mu_host_dev = self.Ktol
        self.nu_host = self.nu
        if me or mi:
            if self.lda0 is None:
                lda = pinv_apply(self._jac_x(x), np.asarray(self.df(x), dtype=np.float64).reshape(n))
                if mi:
                    li = lda[me:]
                    li[li < 0.0] = self.Ktol
            else:
                lda = np.asarray(self.lda0, dtype=np.float64).copy()
        else:
            lda = np.zeros(0)
        self.delta = 0.0
        kkt = self.KKT(x, s, lda)

        if self.lbfgs:                                           # pyipm.py:1633-1637
            zeta, S, Y, SS, L, D, lbfgs_fail = self.lbfgs_init()
            x_old = np.copy(x)
            g = -self.grad(x, s, lda)

        if self.verbosity > 0:
            print('Searching for a feasible local minimizer using L-BFGS to approximate the Hessian.' if self.lbfgs
                  else 'Searching for a feasible local minimizer using the exact Hessian.')
        iter_count = 0
        f_past = float(self.f(x)) if self.Ftol is not None else None
        Ftol_converged = False
        self.signal = 0
        outer = inner = 0

        for outer in range(self.niter):
            if self._kkt_small(kkt, self.Ktol):
                self.signal = 1
                break
            if self.verbosity > 0 and mi:
                print('OUTER ITERATION {}'.format(outer + 1))
            for inner in range(self.miter):
                if self._kkt_small(kkt, max(self.Ktol, self.mu_host)):
                    if not me and not mi:
                        self.signal = 1
                    break
                if self.verbosity > 0:
                    msg = ['* INNER ITERATION {}'.format(inner + 1) if mi else 'ITERATION {}'.format(iter_count + 1)]
                    if self.verbosity > 1:
                        msg.append('f(x) = {}'.format(self.f(x)))
                    if self.verbosity > 2:
                        msg.append('|dL/dx| = {}'.format(np.linalg.norm(kkt[0])))
                        msg.append('|dL/ds| = {}'.format(np.linalg.norm(kkt[1])))
                        msg.append('|ce| = {}'.format(np.linalg.norm(kkt[2])))
                        msg.append('|ci-s| = {}'.format(np.linalg.norm(kkt[3])))
                    print(', '.join(msg))

                if self.lbfgs:                                    # pyipm.py:1702-1713
                    if inner > 0 or outer > 0:
                        g_old = -self.grad(x_old, s, lda)
                        g_new = -self.grad(x, s, lda)
                        zeta, S, Y, SS, L, D, lbfgs_fail = self.lbfgs_update(x_old, x, g_old, g_new, zeta, S, Y, SS,
                                                                             L, D, lbfgs_fail)
                        x_old = np.copy(x)
                        g = np.copy(g_new)
                    dz = self.lbfgs_dir(x, s, lda, g, zeta, S, Y, SS, L, D)
                    if me or mi:
                        dz[n + mi:] = -dz[n + mi:]                # :1723-1725
                else:
                    dz = self.newton_direction(x, s, lda)        # <-- the accelerated hot path

                if me or mi:                                      # merit parameter (pyipm.py:1727-1735)
                    bcg = np.asarray(self.df(x), dtype=np.float64).reshape(n)
                    if mi:
                        bcg = np.concatenate([bcg, -self.mu_host_dev / (s + self.eps)])
                    with np.errstate(divide='ignore', invalid='ignore'):
                        nu_thres = np.dot(bcg, dz[:n + mi]) / (1 - self.rho) / np.sum(np.abs(self._con(x, s)))
                    if self.nu_host < nu_thres:
                        self.nu_host = float(nu_thres)
                if mi:
                    dev = self.backend.step_lengths(self.tau) if hasattr(self.backend, "step_lengths") else None
                    if dev is not None:
                        a_s, a_l = dev
                    else:
                        a_s = self.step(s, dz[n:n + mi])
                        a_l = self.step(lda[me:], dz[n + mi + me:])
                    x, s, lda = self.search(x, s, lda, dz, float(a_s), float(a_l))
                else:
                    x, s, lda = self.search(x, s, lda, dz, 1.0, 1.0)
                iter_count += 1
                kkt = self.KKT(x, s, lda)

                if self.Ftol is not None and not mi and self.signal != -2:
                    f_new = float(self.f(x))
                    if abs(f_past - f_new) <= abs(self.Ftol):
                        self.signal = 2
                        Ftol_converged = True
                        break
                    f_past = f_new
                if self.signal == -2:
                    break
                if inner >= self.miter - 1 and self.verbosity > 0 and mi:
                    print('MAXIMUM INNER ITERATIONS EXCEEDED')

            if self.Ftol is not None and mi and self.signal != -2:
                f_new = float(self.f(x))
                if abs(f_past - f_new) <= abs(self.Ftol):
                    self.signal = 2
                    Ftol_converged = True
                else:
                    f_past = f_new
            if Ftol_converged or self.signal == -2:
                break
            if outer >= self.niter - 1:
                self.signal = -1
                if self.verbosity > 0:
                    print('MAXIMUM OUTER ITERATIONS EXCEEDED' if mi else 'MAXIMUM ITERATIONS EXCEEDED')
                break
            if mi:                                                # barrier update (pyipm.py:1804-1814)
                comp = float(np.dot(s, lda[me:]))
                xi = mi * np.min(s * lda[me:]) / (comp + self.eps)
                mu_new = 0.1 * min(0.05 * (1.0 - xi) / (xi + self.eps), 2.0) ** 3 * comp / mi
                self.mu_host = max(float(mu_new), 0.0)
                self.mu_host_dev = self.mu_host

        self.x, self.s, self.lda, self.kkt = x, s, lda, kkt
        self.fval = self.f(x)
        self.iter_count = iter_count
        if self.verbosity >= 0:
            self._report(kkt, Ftol_converged, outer, inner, iter_count)
        return self.x, self.s, self.lda, self.fval, self.kkt

    def _report(self, kkt, Ftol_converged, outer, inner, iter_count):
        mi = self.nineq
        words = []
        if self.signal == -2:
            words.append('Terminated due to bad direction in backtracking line search')
        elif self._kkt_small(kkt, self.Ktol):
            words.append('Converged to Ktol tolerance')
        elif self.Ftol is not None and Ftol_converged:
            words.append('Converged to Ftol tolerance')
        else:
            words.append('Maximum iterations reached')
            outer, inner = self.niter, 0
        if mi:
            if outer > 1:
                words += ['after {} outer'.format(outer - 1), 'iterations' if outer > 2 else 'iteration', 'and']
            else:
                words.append('after')
            words += ['{} inner'.format(inner), 'iterations' if inner > 1 else 'iteration',
                      '({} total).'.format(iter_count)]
        else:
            words += ['after {}'.format(iter_count), 'iterations.' if iter_count > 1 else 'iteration.']
        print(' '.join(words))
        if self.verbosity > 1:
            msg = ['FINAL: f(x) = {}'.format(self.f(self.x))]
            if self.verbosity > 2:
                msg += ['|dL/dx| = {}'.format(np.linalg.norm(kkt[0])), '|dL/ds| = {}'.format(np.linalg.norm(kkt[1])),
                        '|ce| = {}'.format(np.linalg.norm(kkt[2])), '|ci-s| = {}'.format(np.linalg.norm(kkt[3]))]
            print(', '.join(msg))
