"""Device-resident interior-point loop for the QP family.

    min 1/2 x'Qx + c'x    s.t.  Ax = b ,  Gx - h >= 0

Same algorithm as :class:`pyipm_amd.ipm.IPM` (itself a restatement of ``/root/reference/pyipm.py``:
``:1597-1628`` initial point, ``:1717-1725`` Newton step, ``:1727-1735`` merit parameter,
``:1408-1436`` fraction to the boundary, ``:1438-1565`` backtracking search, ``:1804-1814``
barrier update, ``:958-991`` KKT report) with the provider and the iterate living on the GPU:

* Q, Je = A', Ji = G' are staged ONCE (device pointers handed to the Newton core, no copies);
* df = Qx + c, ce = Ax - b, ci = Gx - h are device GEMVs (the reference evaluates them through
  compiled Aesara functions on the host, ``pyipm.py:855-954``);
* x, s, lambda and dz never leave the device; per line-search trial only the merit value (one
  scalar) crosses PCIe, per iteration the four KKT norms, the step lengths and the merit threshold;
* the fraction-to-the-boundary rule is the closed form ``pyipm_newton_step_lengths``;
* the backtracking search shrinks alpha by tau = 0.995 per trial (``pyipm.py:1534-1548``) — hundreds of merit
  evaluations for one rejected step.  Along the ray the QP's merit function is a closed form in alpha (f quadratic,
  constraints affine), so a trial costs O(me + mi) after three GEMVs per iteration instead of a pass over Q, A, G.

PyTorch supplies the device vectors (plumbing).  The provider's products with the constant blocks -- Q x, A x,
G x and the Jacobian terms of dL/dx, the counterpart of the reference's compiled ``df / ce / ci / grad`` functions
(``pyipm.py:855-954``) -- are the library's own kernels on the blocks the Newton core already holds
(``pyipm_newton_block_products`` / ``_t``: one call gives all three products of a vector), the Newton step
itself is the HIP library.  No CPU fallback: constructing the solver without a GPU raises.

``lbfgs=m`` runs the reference's limited-memory mode (``pyipm.py:1633-1637, 1702-1713``) with the storage
(S, Y) resident too: the direction is ``pyipm_lbfgs_direction`` on device tensors, the constraints are linear so
the Jacobians are staged once and J'J is reused by every direction, and Q may then be given in the factored
form ``("diag+lowrank", d, F)`` = diag(d) + F F' (a dense n x n Q is exactly what L-BFGS is for avoiding).
"""
from __future__ import annotations

import numpy as np

from .newton import NewtonError


class QPDeviceIPM(object):
    def __init__(self, Q, c, A=None, b=None, G=None, h=None, Je=None, Ji=None, x0=None, s0=None, lda0=None,
                 mu=0.2, nu=10.0, rho=0.1, tau=0.995, eta=1.0E-4, beta=0.4, miter=20, niter=10, Xtol=None,
                 Ktol=1.0E-4, Ftol=None, verbosity=1, device=None, nb=256, refine=0, condensed=False,
                 lbfgs=False, lbfgs_zeta=None, warm=True):
        import torch
        from .ipm import HipNewtonBackend
        if not torch.cuda.is_available():
            raise RuntimeError("QPDeviceIPM needs a GPU: the Newton-step core has no CPU fallback")
        self.torch = torch
        dev = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
        self.device = dev

        def dv(a):
            if a is None:
                return None
            if isinstance(a, torch.Tensor):
                return a.to(device=dev, dtype=torch.float64)
            return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64))).to(dev)

        self.c = dv(c)
        n = self.nvar = int(self.c.numel())
        if isinstance(Q, (tuple, list)) and len(Q) == 3 and Q[0] == "diag+lowrank":
            if not lbfgs:
                raise ValueError("a factored Q needs lbfgs=m: the exact-Hessian step stages a dense n x n block")
            self.Q, self.Qd, self.QF = None, dv(Q[1]), dv(Q[2]).contiguous()
        else:
            self.Q, self.Qd, self.QF = dv(Q).contiguous(), None, None
            assert self.Q.shape == (n, n)
        # Jacobians in the reference's layout: Je = dce (n x me), Ji = dci (n x mi)  (pyipm.py:486-501)
        self.Je = dv(Je).contiguous() if Je is not None else (dv(A).t().contiguous() if A is not None else None)
        self.Ji = dv(Ji).contiguous() if Ji is not None else (dv(G).t().contiguous() if G is not None else None)
        self.b, self.h = dv(b), dv(h)
        me = self.neq = 0 if self.Je is None else int(self.Je.shape[1])
        mi = self.nineq = 0 if self.Ji is None else int(self.Ji.shape[1])
        self.x0 = dv(x0) if x0 is not None else torch.zeros(n, dtype=torch.float64, device=dev)
        self.s0, self.lda0 = dv(s0), dv(lda0)
        self.eps = float(np.finfo(np.float64).eps)
        self.mu, self.nu, self.rho, self.tau, self.eta, self.beta = mu, nu, rho, tau, eta, beta
        self.miter, self.niter = int(miter), int(niter)
        self.Xtol = Xtol if Xtol else self.eps
        self.Ktol, self.Ftol = Ktol, Ftol
        self.reg_coef = float(np.sqrt(self.eps))
        self.delta0 = self.reg_coef
        self.verbosity = verbosity
        self.lbfgs = int(lbfgs) if lbfgs else 0
        self.lbfgs_zeta = 1.0 if (lbfgs and lbfgs_zeta is None) else lbfgs_zeta
        if self.lbfgs:
            from .lbfgs import LbfgsCore
            if not 1 <= self.lbfgs <= 31:
                raise ValueError("1 <= lbfgs <= 31")
            self.backend = None
            # the provider's products (df, ce, ci, the J lambda terms) through the library as in the exact-Hessian mode: a
            # provider-only handle -- staged blocks, no KKT storage (pyipm_newton_create_provider); a factored Q stages no
            # d2L block and keeps its own diag + low-rank product
            from .newton import NewtonCore
            self.core = NewtonCore(n, me, mi, device=dev.index, provider_only=True)
            self.core.stage_blocks(self.Q, self.Je, self.Ji)
            self.lb = LbfgsCore(n, me, mi, self.lbfgs + 1, device=dev.index, nb=nb)     # storage reaches lbfgs+1 pairs
            self.lb.stage_jacobian(self.Je, self.Ji)                                    # linear constraints: once
        else:
            self.backend = HipNewtonBackend(n, me, mi, device=dev.index, nb=nb, refine=refine, device_step=True,
                                            condensed=condensed)
            self.core = self.backend.core
            self.core.stage_blocks(self.Q, self.Je, self.Ji)       # constant blocks: device pointers, staged once
        self._pcache = (None, None)
        self._staged_key, self._staged_keep, self._info, self._g = None, None, None, None
        self.warm_seconds = 0.0
        if warm and not self.lbfgs:
            self._warm_up()
        self.trace = None                   # set to [] to record (x, s, lda, mu) at every Newton step
        self.signal = 0
        self.iter_count = 0
        self.timings = {"newton_s": 0.0, "search_s": 0.0, "n_phi": 0, "n_ray": 0, "newton_each_s": []}

    def _warm_up(self):
        """The counterpart of the reference's ``compile()`` (pyipm.py:410-956: seconds of Aesara compilation before the
        first ``solve()``): everything the library builds once per problem shape -- streams, events, the tile lists of the
        update launches, the buffers of the one-launch sweeps -- is built here by ONE Newton step on a harmless point (x = 0,
        s = lda_i = 1, lda_e = 0: the KKT matrix of the staged blocks with Sigma = I), so that the first iteration of
        ``solve()`` costs what the others do (it took 50 ms more at N = 32768).  ``warm=False`` skips it."""
        import time
        torch, n, me, mi = self.torch, self.nvar, self.neq, self.nineq
        t0 = time.perf_counter()
        z = lambda k: torch.zeros(k, dtype=torch.float64, device=self.device)            # noqa: E731
        one = lambda k: torch.ones(k, dtype=torch.float64, device=self.device)           # noqa: E731
        lda = torch.cat([z(me), one(mi)]) if (me or mi) else None
        self.core.stage_vectors(self.c, z(me) if me else None, one(mi) if mi else None, one(mi) if mi else None, lda, mu=self.mu,
                                eps=self.eps)
        try:
            self.core.step(0.0, 0.0)
            if mi:
                self.core.step_lengths(self.tau)
            self.core.merit_info()
        except NewtonError as e:                             # a singular warm-up system is nobody's problem (anything else is a bug: raised)
            self.timings["warmup_error"] = str(e)
        torch.cuda.synchronize(self.device)
        self._staged_key = None
        self.warm_seconds = time.perf_counter() - t0

    # ------------------------------------------------------------------ provider (pyipm.py:855-954 on the device)
    def _products(self, v):
        """(Q v, Je' v, Ji' v): ONE call into the library for all three (cached for the last vector asked about)."""
        key = (id(v), v._version)
        if self._pcache[0] == key:
            return self._pcache[1]
        if self.Q is not None:
            out = self.core.block_products(v)
        else:                                               # L-BFGS mode with a factored Q: its own product, J' v from the library
            _, e, i = self.core.block_products(v, want=(False, True, True))
            out = (self.Qd * v + self.QF @ (self.QF.t() @ v), e, i)
        self._pcache = (key, out)
        self._pkeep = v                                     # keeps id(v) unique while cached
        return out

    def Qx(self, x):
        return self._products(x)[0]

    def f(self, x):
        return float(self.f_dev(x))

    def df(self, x):
        return self.Qx(x) + self.c

    def ce(self, x):
        return self._products(x)[1] - self.b

    def ci(self, x):
        return self._products(x)[2] - self.h

    def _jlam(self, lda):
        """Je lda_e + Ji lda_i (n)."""
        me, mi = self.neq, self.nineq
        return self.core.block_products_t(lda[:me] if me else None, lda[me:] if mi else None)

    def _con(self, x, s):
        parts = []
        if self.neq:
            parts.append(self.ce(x))
        if self.nineq:
            parts.append(self.ci(x) - s)
        return self.torch.cat(parts) if parts else self.torch.zeros(0, dtype=self.torch.float64, device=self.device)

    def grad(self, x, s, lda):
        """KKT residual blocks (pyipm.py:655-668) as a list [gx, gs, ce, ci - s] of device vectors."""
        me, mi = self.neq, self.nineq
        gx = self.df(x)
        if me or mi:
            gx = gx - self._jlam(lda)
        return (gx, (lda[me:] - self.mu_host / (s + self.eps)) if mi else None,
                self.ce(x) if me else None, (self.ci(x) - s) if mi else None)

    # -- the point (x, s, lda) as the library sees it: staged vectors + residual, done once per point -----------------
    def _stage_point(self, x, s, lda):
        """df, ce, ci (the provider's products, one library call), s, lda staged into the handle and g = -grad formed there
        (pyipm_newton_residual) -- ONCE per point and barrier parameter: the KKT report, the merit pieces and the Newton
        step of the next iteration all read this staging (no re-staging inside the loop)."""
        key = (id(x), x._version, id(s), s._version, id(lda), lda._version, float(self.mu_host))
        if self._staged_key == key:
            return
        me, mi = self.neq, self.nineq
        self.core.stage_vectors(self.df(x), self.ce(x) if me else None, self.ci(x) if mi else None, s if mi else None,
                                lda if (me or mi) else None, mu=self.mu_host, eps=self.eps)
        self._g = self.core.residual()                      # device vector, stays on the device
        self._staged_key, self._staged_keep = key, (x, s, lda)
        self._info = None

    def _point_info(self, x, s, lda):
        """pyipm_newton_merit_info for the staged point without a direction (KKT norms, ||c||_1, sum log s, s'lda, min)."""
        self._stage_point(x, s, lda)
        if self._info is None:
            self._info = self.core.merit_info()
        return self._info

    def KKT(self, x, s, lda):
        """Norms of the four first-order blocks, slack block scaled by s (pyipm.py:958-991): device reductions of the
        library (k_merit_info), one D2H."""
        q = self._point_info(x, s, lda)
        return (q["kkt_x"], q["kkt_s"] if self.nineq else 0.0, q["kkt_ce"] if self.neq else 0.0,
                q["kkt_ci"] if self.nineq else 0.0)

    def f_dev(self, x):
        """f(x) = x'Qx/2 + c'x from the cached product (pyipm_newton_dots: one launch, one D2H)."""
        d = self.core.dots([(x, self.Qx(x)), (self.c, x)])
        return 0.5 * d[0] + d[1]

    def phi(self, x, s, lda=None):
        """Merit function (pyipm.py:670-721) at an ARBITRARY point: its vectors are staged (the staged point changes --
        search() restages its own before it goes on) and reduced by the library."""
        me, mi = self.neq, self.nineq
        if not hasattr(self, "mu_host"):                     # (called before solve(): the initial barrier parameter)
            self.mu_host, self.nu_host = (self.mu if mi else self.Ktol), self.nu
        if lda is None:
            # phi itself does not depend on lda; the staging does (comp_sum / kkt entries of merit_info): the multipliers of
            # the point staged last when there is one, zeros before any point was staged
            lda = self._staged_keep[2] if self._staged_keep is not None else \
                self.torch.zeros(me + mi, dtype=self.torch.float64, device=self.device)
        self.core.stage_vectors(self.df(x), self.ce(x) if me else None, self.ci(x) if mi else None, s if mi else None,
                                lda if (me or mi) else None, mu=self.mu_host, eps=self.eps)
        self._staged_key = None
        q = self.core.merit_info()
        self.timings["n_phi"] += 1
        self._last_con_l1 = (q["ce_l1"] if me else 0.0) + (q["cis_l1"] if mi else 0.0)
        v = self.f_dev(x)
        if me or mi:
            v = v + self.nu_host * self._last_con_l1
        if mi:
            v = v - self.mu_host * q["sum_log_s"]
        return v

    def _quad(self, dx):
        """dx'Q dx when Q is not a staged dense block (L-BFGS mode with a factored Q); None lets the library form it."""
        if self.Q is not None:
            return None
        return self.core.dots([(dx, self.Qd * dx + self.QF @ (self.QF.t() @ dx))])[0]

    def step(self, x, dx):
        """Largest alpha in [0,1] with x + alpha dx >= (1 - tau) x: closed form of the golden-section
        search of pyipm.py:1408-1436."""
        torch = self.torch
        neg = dx < 0
        if not bool(neg.any()):
            return 1.0
        return min(1.0, float((-self.tau * x[neg] / dx[neg]).min()))

    def _restoration(self, x0, c_new):
        """Minimum-norm feasibility-restoration direction (pyipm.py:1466-1477, 1518-1529): -pinv(At) c for the constraint
        Jacobian At = [[Je', 0], [Ji', -I]] over (x, s).  At At' = J'J + diag(0, I) is the Gram matrix of the library's
        L-BFGS direction with zeta = 1 and Sigma = 1 (include/pyipm_lbfgs.h: G = J'J / zeta + diag(0, 1 / Sigma)), and with an
        empty storage that direction is Zg = [inv(A)(g1 - B y); y], y = inv(G)(B'inv(A) g1 - g2): for g1 = 0, g2 = -c its (x, s)
        part IS -At'(At At')^-1 c.  So the step comes from the library's own kernels (split-K MFMA Gram launch -- cached while
        the Jacobians do not change --, block LDL', two passes over J); round 4 went through torch.linalg.cholesky_ex and a
        rocBLAS product.  Guard: a factor with rejected / negative pivots, or a step that does not satisfy At z = -c, takes the
        SVD as before (rank-deficient constraints)."""
        torch = self.torch
        n, me, mi = self.nvar, self.neq, self.nineq
        f64, dev = torch.float64, self.device
        try:
            core = self._restoration_core()
            g = torch.cat([torch.zeros(n + mi, dtype=f64, device=dev), -c_new])
            s1 = torch.full((mi,), 1.0 - self.eps, dtype=f64, device=dev) if mi else None       # (s + eps == 1 exactly)
            l1 = torch.cat([torch.zeros(me, dtype=f64, device=dev), torch.ones(mi, dtype=f64, device=dev)])
            dz, st = core.direction(g, s1, l1, 1.0, None, None, None, None, None, reg=0.0, eps=self.eps, flip=False)
            z = dz[:n + mi]
            if st["n_neg"] == 0 and st["n_zero"] == 0 and not st["regularised"] and bool(torch.isfinite(z).all()):
                _, je, ji = self.core.block_products(z[:n].contiguous(), want=(False, True, True))
                parts = ([je] if me else []) + ([ji - z[n:]] if mi else [])
                res = torch.cat(parts) + c_new                                               # At z + c
                if float(res.norm()) <= 1.0e-9 * max(float(c_new.norm()), 1.0e-300):
                    return z
        except (NewtonError, torch.cuda.OutOfMemoryError) as e:      # (no memory for the Gram matrix, a rank-deficient Jacobian: the SVD
            self.timings.setdefault("fallbacks", []).append("restoration: %s" % e)   #  path below; programming errors are not swallowed)
        cols = [m for m in (self.Je, self.Ji) if m is not None]
        top = torch.cat(cols, dim=1)
        if mi:
            bottom = torch.cat([torch.zeros((mi, me), dtype=f64, device=dev), -torch.eye(mi, dtype=f64, device=dev)], dim=1)
            top = torch.cat([top, bottom], dim=0)
        return -self._pinv_svd(top.t().contiguous(), c_new)

    def close(self):
        """Release the library handles this solver owns (the restoration step's Gram workspace is (me + mi)^2 doubles)."""
        rc = getattr(self, "_rcore", None)
        if rc is not None:
            rc.close()
            self._rcore = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _restoration_core(self):
        """The L-BFGS direction handle the restoration step runs on: the limited-memory mode's own (Jacobians staged, J'J cached),
        or one created on first use and kept for the solve."""
        if self.lbfgs:
            return self.lb
        if getattr(self, "_rcore", None) is None:
            from .lbfgs import LbfgsCore
            self._rcore = LbfgsCore(self.nvar, self.neq, self.nineq, 1, device=self.device.index)
            self._rcore.stage_jacobian(self.Je, self.Ji)
        return self._rcore

    RAY_BATCH = 64                       # backtracking candidates alpha tau^k evaluated per launch of k_merit_ray

    def search(self, x0, s0, lda0, dz, alpha_smax, alpha_lmax, info=None):
        """Backtracking Armijo search with the optional second-order correction (pyipm.py:1438-1565).  The merit function
        and its slope come from the library's device reductions over the staged point (pyipm_newton_merit_info), the trial
        values from ONE launch per batch of candidates alpha tau^k (pyipm_newton_merit_ray): phi(x0 + a dx, s0 + a ds) -
        phi(x0, s0) in closed form, as a difference (see kernels_merit.hpp)."""
        torch = self.torch
        n, me, mi = self.nvar, self.neq, self.nineq
        dx = dz[:n]
        ds = dz[n:n + mi]
        dl = dz[n + mi:]
        if not (me or mi):
            alpha_lmax = 0.0
        self._stage_point(x0, s0, lda0)
        q = info if info is not None else self.core.merit_info(dz=dz)
        con_l1 = (q["ce_l1"] if me else 0.0) + (q["cis_l1"] if mi else 0.0)
        phi0 = self.f_dev(x0)
        if me or mi:
            phi0 = phi0 + self.nu_host * con_l1
        if mi:
            phi0 = phi0 - self.mu_host * q["sum_log_s"]
        dphi0 = q["df_dx"]
        if me or mi:
            dphi0 = dphi0 - self.nu_host * con_l1
        if mi:
            dphi0 = dphi0 - self.mu_host * q["ds_over_s"]
        self.timings["n_phi"] += 1
        armijo = lambda a: phi0 + a * self.eta * dphi0     # noqa: E731
        quad = self._quad(dx)

        def trials(alphas):
            """> 0: rejected.  Rounded to the merit function's own precision, as the reference's two evaluations are: a
            change below ulp(phi0) compares equal and is accepted, which is how its last tiny steps pass (pyipm.py:1454-1459)."""
            self._stage_point(x0, s0, lda0)                # (the correction below may have staged another point)
            d = self.core.merit_ray(alphas, self.nu_host, self.mu_host, dz=dz, quad=quad)
            self.timings["n_phi"] += len(alphas)
            self.timings["n_ray"] = self.timings.get("n_ray", 0) + 1
            return [(phi0 + dk) - armijo(ak) for ak, dk in zip(alphas, d)]

        corrected, alpha_corr, dz_p = False, 1.0, None
        # the first batch already holds the backtracking candidates behind alpha_smax: one launch for most searches
        # (the reference multiplies step by step, alpha *= tau: the same sequence of roundings)
        cand, cand_l = [alpha_smax], [alpha_lmax]
        for k in range(1, self.RAY_BATCH):
            cand.append(cand[-1] * self.tau)
            cand_l.append(cand_l[-1] * self.tau)
        tr = trials(cand)
        if tr[0] > 0.0:
            if me or mi:
                c_old_l1 = con_l1
                xt, st = x0 + alpha_smax * dx, (s0 + alpha_smax * ds if mi else s0)
                self.phi(xt, st)
                if self._last_con_l1 > c_old_l1:
                    dz_p = self._restoration(x0, self._con(xt, st))
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
                ndx = q["dx_norm"]
                nds = q["ds_norm"] if mi else 0.0
                k = 1                                        # cand[k] = alpha_smax tau^k is the candidate on trial
                while True:
                    if k >= len(cand):                       # next batch, continuing the same sequence of roundings
                        a0, l0 = cand[-1], cand_l[-1]
                        cand, cand_l = [], []
                        for _ in range(self.RAY_BATCH):
                            a0 *= self.tau; l0 *= self.tau
                            cand.append(a0); cand_l.append(l0)
                        tr = trials(cand)
                        k = 0
                    alpha_smax, alpha_lmax = cand[k], cand_l[k]
                    if not tr[k] > 0.0:
                        break
                    size = np.sqrt((alpha_smax * ndx) ** 2 + (alpha_lmax * nds) ** 2) if mi else alpha_smax * ndx
                    if size < self.eps:
                        if self.verbosity > 2:
                            print('Search direction is unreliable to machine precision.')
                        self.signal = -2
                        return x0, s0, lda0
                    k += 1
        if corrected:
            x = x0 + alpha_corr * (alpha_smax * dx + dz_p[:n])
            s = s0 + alpha_corr * (alpha_smax * ds + dz_p[n:]) if mi else s0.clone()
        else:
            x = x0 + alpha_smax * dx
            s = s0 + alpha_smax * ds if mi else s0.clone()
        lda = lda0 + alpha_lmax * dl if (me or mi) else lda0.clone()
        return x, s, lda

    # ------------------------------------------------------------------ the Newton step (hot path)
    def newton_direction(self, x, s, lda):
        """pyipm.py:1717-1725 on the device: the blocks are resident (staged once), the vectors and g = -grad of this point
        are already staged (_stage_point: the KKT report of the previous iteration did it)."""
        if self.trace is not None:
            self.trace.append((x.cpu().numpy(), s.cpu().numpy(), lda.cpu().numpy(), float(self.mu_host)))
        self._stage_point(x, s, lda)
        dz, self.delta, self.last_stats = self.backend.direction(
            None, None, None, None, None, None, None, None, self.mu_host, self.delta, self.mu_host, self.eta,
            self.beta, self.reg_coef, self.delta0, self.eps, as_tensor=True, staged=True, g=self._g)
        return dz

    # ------------------------------------------------------------------ L-BFGS mode (storage on the device)
    def _gvec(self, x, s, lda):
        """-grad as ONE device vector (pyipm.py:1637, 1705-1706): the library's residual of the staged point."""
        self._stage_point(x, s, lda)
        return self._g[:self.nvar + 2 * self.nineq + self.neq].clone()

    def _lbfgs_init(self):
        t, n, dev = self.torch, self.nvar, self.device
        e = lambda r, c: t.zeros((r, c), dtype=t.float64, device=dev)       # noqa: E731
        z = np.zeros((0, 0))
        return float(self.lbfgs_zeta), e(n, 0), e(n, 0), z, z.copy(), z.copy(), 0

    def _lbfgs_update(self, x_old, x_new, g_old, g_new, zeta, S, Y, SS, L, D, fail):
        """pyipm.py:1282-1371 with S, Y on the device; only the O(m) inner products cross PCIe."""
        t, n = self.torch, self.nvar
        con = bool(self.neq or self.nineq)
        dx = x_new - x_old
        dg = g_old[:n] - g_new[:n]
        k = S.shape[1]
        drop = k > self.lbfgs
        Sn = t.cat([S[:, 1:] if drop else S, dx[:, None]], dim=1)
        Yn = t.cat([Y[:, 1:] if drop else Y, dg[:, None]], dim=1)
        prods = t.cat([Sn.t() @ dx if con else Yn.t() @ dg, dx @ Yn if con else Sn.t() @ dg,
                       t.stack([t.dot(dg, dx), t.dot(dx, dx) if con else t.dot(dg, dg)])]).tolist()     # ONE sync
        kk = Sn.shape[1]
        inner, cross, curv, den = np.array(prods[:kk]), np.array(prods[kk:2 * kk]), prods[-2], prods[-1]
        zeta_new = curv / (den + self.eps)
        root = np.sqrt(self.eps)
        if curv > root and zeta_new > root:
            if drop:
                SS, L, D = SS[1:, 1:], L[1:, 1:], D[1:, 1:]
            SS, L, D = (np.pad(Mx, ((0, 1), (0, 1))) for Mx in (SS, L, D))
            SS[:, -1] = SS[-1, :] = inner
            if con:
                L[-1, :] = cross
                L[-1, -1] = 0.0
            else:
                L[:, -1] = cross
            D[-1, -1] = curv
            zeta, S, Y, fail = zeta_new, Sn.contiguous(), Yn.contiguous(), 0
        else:
            fail += 1
        if fail > self.lbfgs and S.shape[1] > 0:
            return self._lbfgs_init()
        return zeta, S, Y, SS, L, D, fail

    @staticmethod
    def _pinv_svd(J, g):
        """pinv(J) @ g through an SVD: the last resort for rank-deficient or badly conditioned Jacobians."""
        import torch
        if J.numel() <= (1 << 22):
            # small enough for the host: LAPACK's SVD (the device SVD loses the small singular values of such a matrix -- 1.28e-9
            # for 1.00e-9 on a 30 x 70 example -- and pinv is all about those)
            return torch.from_numpy(np.linalg.pinv(J.cpu().numpy()) @ g.cpu().numpy()).to(J.device)
        return torch.linalg.pinv(J) @ g

    @staticmethod
    def _pinv_apply(J, g, matvec=None, rmatvec=None, info=None):
        """pinv(J) @ g, the reference's first multiplier estimate (pyipm.py:726-730), for J (n x m) on the device.  When J has full
        rank and is not badly conditioned the same vector comes from the normal equations -- (J'J)^-1 J'g for a tall J,
        J'(J J')^-1 g for a wide one -- formed and solved by the LIBRARY'S OWN kernels (round 5; rounds 3-4: torch.linalg.cholesky_ex
        and a rocBLAS product): the L-BFGS direction of include/pyipm_lbfgs.h with an empty storage, zeta = 1 and every constraint
        an equality is  y = (J'J)^-1 (J'g1 - g2)  -- a split-K MFMA Gram launch, the block LDL' of the Newton core, two passes
        over J -- and its x part (g1 - J y).  One refinement step on the least-squares residual follows.  Accepted only if the
        factorisation found G positive definite with pivots spanning less than six decades AND the result satisfies the normal
        equations to working precision (a pivot spread does not bound cond(J'J)); otherwise the SVD as before.
        matvec(lam) = J lam / rmatvec(v) = J'v: the caller's own products (QPDeviceIPM: the library's block products); default:
        torch's."""
        import torch
        from .lbfgs import LbfgsCore
        n, m = J.shape
        f64, dev = torch.float64, J.device
        if matvec is None:
            matvec = lambda v: J @ v                    # noqa: E731
        if rmatvec is None:
            rmatvec = lambda v: J.t() @ v               # noqa: E731
        z = lambda k: torch.zeros(k, dtype=f64, device=dev)     # noqa: E731
        wide = m >= n
        core = None
        try:
            if wide:                                    # Gram over the other dimension: K = J' (m x n) plays the Jacobian
                core = LbfgsCore(m, n, 0, 1, device=dev.index)
                core.stage_jacobian(J.t().contiguous(), None)
            else:
                core = LbfgsCore(n, m, 0, 1, device=dev.index)
                core.stage_jacobian(J.contiguous(), None)
            stats = []

            def apply(r):                               # pinv(J) r through the factor of the Gram matrix
                if wide:                                # y = (J J')^-1 r from g1 = 0, g2 = -r; the x part is -K y = -J'y
                    dz, st = core.direction(torch.cat([z(m), -r]), None, z(n), 1.0, None, None, None, None, None, reg=0.0)
                    stats.append(st)
                    return -dz[:m]
                dz, st = core.direction(torch.cat([r, z(m)]), None, z(m), 1.0, None, None, None, None, None, reg=0.0)
                stats.append(st)
                return dz[n:].clone()

            lam = apply(g)
            st = stats[0]
            ok = (st["n_neg"] == 0 and st["n_zero"] == 0 and not st["regularised"] and st["d_max"] > 0.0 and
                  st["d_min"] > 1.0e-6 * st["d_max"] and bool(torch.isfinite(lam).all()))
            if ok:
                # one refinement step on the least-squares residual (wide: J lam = g exactly; tall: J'(g - J lam) = 0)
                lam = lam + apply(g - matvec(lam))
                r = g - matvec(lam)
                if wide:
                    ok = float(r.norm()) <= 1.0e-11 * max(float(g.norm()), 1.0e-300)
                else:
                    ok = float(rmatvec(r).norm()) <= 1.0e-11 * max(float(J.norm()) * float(g.norm()), 1.0e-300)
                if ok and bool(torch.isfinite(lam).all()):
                    if info is not None:
                        info["path"] = "normal equations (library Gram + block LDL')"
                    return lam
        except (NewtonError, torch.cuda.OutOfMemoryError) as e:     # (no memory for the Gram matrix: the SVD path needs less)
            if info is not None:
                info["fallback_reason"] = str(e)
        finally:
            if core is not None:
                core.close()
        if info is not None:
            info["path"] = "svd"
        return QPDeviceIPM._pinv_svd(J, g)

    def _small(self, kkt, tol):
        return all(k <= tol for k in kkt)

    # ------------------------------------------------------------------ driver (pyipm.py:1567-1863)
    def solve(self):
        import time
        torch = self.torch
        n, me, mi = self.nvar, self.neq, self.nineq
        x = self.x0.clone()
        if mi:
            s = torch.clamp(self.ci(x), min=self.Ktol) if self.s0 is None else self.s0.clone()
            self.mu_host = self.mu
        else:
            s = torch.zeros(0, dtype=torch.float64, device=self.device)
            self.mu_host = self.Ktol
        self.nu_host = self.nu
        if me or mi:
            if self.lda0 is None:
                t_init = time.perf_counter()
                J = torch.cat([m for m in (self.Je, self.Ji) if m is not None], dim=1)
                # (the residual checks of the estimate through the library's own block products, as everything else in the loop)
                info = {}
                lda = self._pinv_apply(J, self.df(x), matvec=self._jlam,
                                       rmatvec=lambda v: torch.cat([t for t in self.core.block_products(v, want=(False, True, True))[1:]
                                                                    if t is not None]), info=info)
                del J
                torch.cuda.synchronize(self.device)
                self.timings["init_multipliers_s"] = time.perf_counter() - t_init
                self.timings["init_multipliers_path"] = info.get("path")
                if mi:
                    li = lda[me:]
                    li[li < 0.0] = self.Ktol
            else:
                lda = self.lda0.clone()
        else:
            lda = torch.zeros(0, dtype=torch.float64, device=self.device)
        self.delta = 0.0
        kkt = self.KKT(x, s, lda)
        if self.lbfgs:
            zeta, S, Y, SS, L, D, lb_fail = self._lbfgs_init()
            x_old, g = x.clone(), self._gvec(x, s, lda)
        if self.verbosity > 0:
            print('Searching for a feasible local minimizer using L-BFGS to approximate the Hessian.' if self.lbfgs
                  else 'Searching for a feasible local minimizer using the exact Hessian.')
        iter_count = 0
        f_past = self.f(x) if self.Ftol is not None else None
        Ftol_converged = False
        self.signal = 0
        outer = inner = 0
        for outer in range(self.niter):
            if self._small(kkt, self.Ktol):
                self.signal = 1
                break
            if self.verbosity > 0 and mi:
                print('OUTER ITERATION {}'.format(outer + 1))
            for inner in range(self.miter):
                if self._small(kkt, max(self.Ktol, self.mu_host)):
                    if not me and not mi:
                        self.signal = 1
                    break
                if self.verbosity > 0:
                    msg = ['* INNER ITERATION {}'.format(inner + 1) if mi else 'ITERATION {}'.format(iter_count + 1)]
                    if self.verbosity > 1:
                        msg.append('f(x) = {}'.format(self.f(x)))
                    if self.verbosity > 2:
                        msg += ['|dL/dx| = {}'.format(kkt[0]), '|dL/ds| = {}'.format(kkt[1]),
                                '|ce| = {}'.format(kkt[2]), '|ci-s| = {}'.format(kkt[3])]
                    print(', '.join(msg))
                t0 = time.perf_counter()
                if self.lbfgs:                                    # pyipm.py:1702-1713, 1723-1725
                    if inner > 0 or outer > 0:
                        g_old, g_new = self._gvec(x_old, s, lda), self._gvec(x, s, lda)
                        zeta, S, Y, SS, L, D, lb_fail = self._lbfgs_update(x_old, x, g_old, g_new, zeta, S, Y, SS, L, D,
                                                                           lb_fail)
                        x_old, g = x.clone(), g_new
                    reg = self.reg_coef * self.eta * (self.mu_host ** self.beta)
                    dz, self.last_stats = self.lb.direction(g, s if mi else None, lda if (me or mi) else None, zeta,
                                                            S, Y, SS, L, D, reg=reg, eps=self.eps, flip=True)
                    if mi:
                        a_s, a_l = self.step(s, dz[n:n + mi]), self.step(lda[me:], dz[n + mi + me:])
                else:
                    dz = self.newton_direction(x, s, lda)        # <-- the accelerated hot path
                    if mi:
                        a_s, a_l = self.backend.step_lengths(self.tau)
                torch.cuda.synchronize(self.device)
                t1 = time.perf_counter()
                self._stage_point(x, s, lda)                      # (L-BFGS mode: g of the current point was staged last)
                q = self.core.merit_info(dz=dz)                   # ||c||_1, df.dx, sum ds/(s+eps), sum log s, |dx|, |ds|: one D2H
                if me or mi:                                      # merit parameter (pyipm.py:1727-1735)
                    con_l1 = (q["ce_l1"] if me else 0.0) + (q["cis_l1"] if mi else 0.0)
                    den = (1 - self.rho) * con_l1
                    num = q["df_dx"] - (self.mu_host * q["ds_over_s"] if mi else 0.0)
                    with np.errstate(divide='ignore', invalid='ignore'):
                        nu_thres = np.float64(num) / np.float64(den)
                    if self.nu_host < nu_thres:
                        self.nu_host = float(nu_thres)
                if mi:
                    x, s, lda = self.search(x, s, lda, dz, float(a_s), float(a_l), info=q)
                else:
                    x, s, lda = self.search(x, s, lda, dz, 1.0, 1.0, info=q)
                iter_count += 1
                kkt = self.KKT(x, s, lda)
                t2 = time.perf_counter()
                self.timings["newton_s"] += t1 - t0
                self.timings["newton_each_s"].append(t1 - t0)
                self.timings["search_s"] += t2 - t1
                if self.Ftol is not None and not mi and self.signal != -2:
                    f_new = self.f(x)
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
                f_new = self.f(x)
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
                q = self._point_info(x, s, lda)                   # s'lda_i and min s lda_i of the staged point (k_merit_info)
                comp, mn = q["comp_sum"], q["comp_min"]
                xi = mi * mn / (comp + self.eps)
                mu_new = 0.1 * min(0.05 * (1.0 - xi) / (xi + self.eps), 2.0) ** 3 * comp / mi
                self.mu_host = max(float(mu_new), 0.0)
        self.x, self.s, self.lda, self.kkt = x, s, lda, kkt
        self.fval = self.f(x)
        self.iter_count = iter_count
        if self.verbosity >= 0:
            words = ('Terminated due to bad direction in backtracking line search' if self.signal == -2 else
                     'Converged to Ktol tolerance' if self._small(kkt, self.Ktol) else
                     'Converged to Ftol tolerance' if Ftol_converged else 'Maximum iterations reached')
            print('{} ({} total iterations).'.format(words, iter_count))
        return self.x, self.s, self.lda, self.fval, self.kkt
