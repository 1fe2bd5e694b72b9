#!/usr/bin/env python
"""Generate tests/golden/*.npz by driving the UNMODIFIED reference.  TEST INFRASTRUCTURE ONLY.

Runs only in the build container: it imports ``/root/reference/pyipm.py`` by path
(never copied into this repo, never shipped to the GPU box) on top of the
test-only ``aesara`` stand-in in ``oracle/aesara_standin`` and records what the
reference computes at the Newton-step seam (pyipm.py:1717-1725):

  x, s, lda, mu, delta_in  ->  g = -grad,  H = hess,  Hc = reghess(H), delta_out,
  dz_raw = sym_solve_cmp(Hc, g),  dz = flipped

All nine inputs are handed over as "precompiled" Functions (the input state of
pyipm.py:426-440 that needs no autodiff), so the reference's own NumPy KKT
assembly (:768-814), ``reghess`` (:1373-1406), line search and loop run verbatim.

    python oracle/make_golden.py            # writes tests/golden/*.npz (exact-Hessian seam)
    python oracle/make_golden.py --lbfgs    # writes tests/golden/lbfgs_*.npz (L-BFGS seam, pyipm.py:1713)
"""
from __future__ import annotations

import io
import os
import sys
import contextlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REFERENCE = "/root/reference"
sys.path.insert(0, os.path.join(HERE, "aesara_standin"))
sys.path.insert(0, REFERENCE)
sys.path.insert(0, ROOT)

import aesara  # noqa: E402  (the stand-in)
import aesara.tensor as T  # noqa: E402
import pyipm as ref  # noqa: E402  (the unmodified reference)

from pyipm_amd.problems import example_problem, unit_test_x0, make_qp, qp_callables  # noqa: E402

assert ref.__file__.startswith(REFERENCE), ref.__file__
GOLD = os.path.join(ROOT, "tests", "golden")


class TracedIPM(ref.IPM):
    """Reference IPM with recording hooks at the Newton-step seam (no logic changed)."""

    def compile(self, *a, **k):
        ref.IPM.compile(self, *a, **k)
        self.trace = []
        self._cur = None
        if self.lbfgs:
            return                      # the L-BFGS seam is recorded by lbfgs_dir below
        hess0, solve0 = self.hess, self.sym_solve_cmp

        def hess(x, s, lda):
            H = hess0(x, s, lda)
            self._cur = {"x": np.array(x), "s": np.array(s), "lda": np.array(lda),
                         "mu": float(self.mu_dev.get_value()), "mu_host": float(self.mu_host),
                         "g": -np.array(self.grad(x, s, lda)), "H": np.array(H)}
            return H

        def solve(Mx, b):
            out = solve0(Mx, b)
            if self._cur is not None and "Hc" in self._cur and "dz_raw" not in self._cur:
                self._cur["dz_raw"] = np.array(out).reshape(-1)
                self.trace.append(self._cur)
                self._cur = None
            return out

        self.hess, self.sym_solve_cmp = hess, solve

    def lbfgs_dir(self, x, s, lda, g, zeta, S, Y, SS, L, D):
        """Record the L-BFGS seam (pyipm.py:1713): inputs of lbfgs_dir and the RAW direction it returns."""
        dz = ref.IPM.lbfgs_dir(self, x, s, lda, g, zeta, S, Y, SS, L, D)
        self.trace.append({"x": np.array(x), "s": np.array(s), "lda": np.array(lda), "g": np.array(g),
                           "zeta": float(zeta), "S": np.array(S), "Y": np.array(Y), "SS": np.array(SS),
                           "L": np.array(L), "D": np.array(D), "mu": float(self.mu_dev.get_value()),
                           "mu_host": float(self.mu_host), "dz_raw": np.array(dz)})
        return dz

    def reghess(self, Hc):
        rec = self._cur
        if rec is not None:
            rec["delta_in"] = float(self.delta)
        out = ref.IPM.reghess(self, Hc)
        if rec is not None:
            rec["Hc"] = np.array(out)
            rec["delta_out"] = float(self.delta)
        return out


def as_functions(prob, x_dev, lda_dev):
    """Wrap NumPy callables as stand-in ``Function`` objects ("precompiled" state)."""
    out = {}
    for key in ("f", "df", "d2f", "ce", "dce", "ci", "dci"):
        fn = prob.get(key)
        out[key] = None if fn is None else aesara.function([x_dev], aesara.wrap(fn, x_dev))
    for key in ("d2ce", "d2ci"):
        fn = prob.get(key)
        out[key] = None if fn is None else aesara.function([x_dev, lda_dev], aesara.wrap(fn, x_dev, lda_dev))
    return out


def build(prob, x0, **kw):
    x_dev = T.vector("x_dev")
    lda_dev = T.vector("lda_dev")
    fns = as_functions(prob, x_dev, lda_dev)
    return TracedIPM(x0=np.array(x0, dtype=np.float64), x_dev=x_dev, lambda_dev=lda_dev,
                     f=fns["f"], df=fns["df"], d2f=fns["d2f"], ce=fns["ce"], dce=fns["dce"],
                     d2ce=fns["d2ce"], ci=fns["ci"], dci=fns["dci"], d2ci=fns["d2ci"], **kw)


def flipped(dz_raw, n, mi, me):
    dz = np.array(dz_raw)
    if me or mi:
        dz[n + mi:] = -dz[n + mi:]
    return dz


def pack_trace(p):
    n, me, mi = p.nvar, p.neq, p.nineq
    keys = ("x", "s", "lda", "g", "H", "Hc", "dz_raw")
    d = {"n_iter": np.int64(len(p.trace))}
    for k in keys:
        d["it_" + k] = np.stack([t[k] for t in p.trace]) if p.trace else np.zeros((0,))
    for k in ("mu", "mu_host", "delta_in", "delta_out"):
        d["it_" + k] = np.array([t[k] for t in p.trace])
    d["it_dz"] = np.stack([flipped(t["dz_raw"], n, mi, me) for t in p.trace]) if p.trace else np.zeros((0,))
    return d


def pack_lbfgs_trace(p, memory):
    """Variable-size storage zero-padded to memory+1 pairs (the reference lets it grow that far, :1300)."""
    n, cap = p.nvar, memory + 1
    d = {"n_iter": np.int64(len(p.trace)), "it_m": np.array([t["S"].shape[1] for t in p.trace], dtype=np.int64)}
    for k in ("x", "s", "lda", "g", "dz_raw"):
        d["it_" + k] = np.stack([t[k] for t in p.trace])
    for k in ("zeta", "mu", "mu_host"):
        d["it_" + k] = np.array([t[k] for t in p.trace])
    for k in ("S", "Y"):
        d["it_" + k] = np.stack([np.pad(t[k], ((0, 0), (0, cap - t[k].shape[1]))) for t in p.trace])
    for k in ("SS", "L", "D"):
        d["it_" + k] = np.stack([np.pad(t[k], ((0, cap - t[k].shape[0]), (0, cap - t[k].shape[1]))) for t in p.trace])
    return d


def run_example_lbfgs(k, x0, memory, **kw):
    prob = dict(example_problem(k))
    for key in ("d2f", "d2ce", "d2ci"):
        prob[key] = None                                  # L-BFGS takes no second derivatives (:478-562)
    p = build(prob, x0, lbfgs=memory, **kw)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), np.errstate(all="ignore"):
        x, s, lda, fval, kkt = p.solve()
    d = pack_lbfgs_trace(p, memory)
    d.update(problem=np.int64(k), x0=np.array(x0), x=x, s=s, lda=lda, fval=np.float64(fval),
             signal=np.int64(p.signal), nvar=np.int64(p.nvar), neq=np.int64(p.neq), nineq=np.int64(p.nineq),
             memory=np.int64(memory), transcript=np.array(buf.getvalue()))
    return d, p


def lbfgs_direction_fixture(n, me, mi, m, seed, rank_deficient=False):
    """One call of the reference's lbfgs_dir on QP-shaped data with a storage built by the reference's own
    lbfgs_update from random displacement pairs of positive curvature."""
    rng = np.random.default_rng(seed)
    qp = make_qp(n, me, mi, seed)
    if rank_deficient and me >= 2:
        qp["Je"][:, me - 1] = qp["Je"][:, 0]              # duplicated equality gradient -> rcond = 0 (:1108-1113)
    Je, Ji = qp["Je"], qp["Ji"]
    prob = {"nvar": n, "neq": me, "nineq": mi, "f": lambda x: 0.0, "df": lambda x: np.zeros(n), "d2f": None,
            "ce": (lambda x: np.zeros(me)) if me else None, "dce": (lambda x: Je) if me else None, "d2ce": None,
            "ci": (lambda x: np.ones(mi)) if mi else None, "dci": (lambda x: Ji) if mi else None, "d2ci": None}
    p = build(prob, np.zeros(n), lbfgs=max(m, 1), verbosity=-1)
    p.nvar = n
    p.compile()
    p.mu_host = 0.2
    p.mu_dev.set_value(np.float64(0.2))
    zeta, S, Y, SS, L, D, fail = p.lbfgs_init()
    Mq = rng.standard_normal((n, n)) / np.sqrt(n)
    Qh = Mq @ Mq.T + 0.5 * np.eye(n)                     # curvature model: dg = Qh dx  => dx'dg > 0
    pad = n + 2 * mi + me - n
    x_old = rng.standard_normal(n)
    for _ in range(m):
        x_new = x_old + rng.standard_normal(n) / np.sqrt(n)
        g_old = np.concatenate([-(Qh @ x_old), np.zeros(pad)])
        g_new = np.concatenate([-(Qh @ x_new), np.zeros(pad)])
        zeta, S, Y, SS, L, D, fail = p.lbfgs_update(x_old, x_new, g_old, g_new, zeta, S, Y, SS, L, D, fail)
        x_old = x_new
    g = rng.standard_normal(n + 2 * mi + me)
    s = qp["s"] if mi else np.array([])
    lda = qp["lam"] if (me or mi) else np.array([])
    dz = p.lbfgs_dir(np.zeros(n), s, lda, g, zeta, S, Y, SS, L, D)
    out = {"n": np.int64(n), "me": np.int64(me), "mi": np.int64(mi), "m": np.int64(S.shape[1]), "seed": np.int64(seed),
           "g": g, "s": s, "lda": lda, "zeta": np.float64(zeta), "S": S, "Y": Y, "SS": SS, "L": L, "D": D,
           "mu": np.float64(0.2), "eta": np.float64(p.eta), "beta": np.float64(p.beta),
           "reg_coef": np.float64(p.reg_coef), "dz_raw": np.array(dz),
           "rank_deficient": np.int64(bool(rank_deficient and me >= 2))}
    if me:
        out["Je"] = Je
    if mi:
        out["Ji"] = Ji
    return out


def main_lbfgs():
    """tests/golden/lbfgs_*.npz."""
    os.makedirs(GOLD, exist_ok=True)
    x0s = unit_test_x0()
    for k in range(1, 11):                                # unit-test setting: lbfgs=4, Ftol=1e-8 (unit_tests.py:49-50)
        d, p = run_example_lbfgs(k, x0s[k], 4, Ftol=1.0e-8, verbosity=-1)
        np.savez_compressed(os.path.join(GOLD, "lbfgs_trace_p%02d.npz" % k), **d)
        print("lbfgs p%-2d iters=%-3d signal=%2d x=%s" % (k, int(d["n_iter"]), int(d["signal"]), d["x"]))
    for (n, me, mi, m, seed, rd) in [(40, 0, 0, 3, 0, False), (64, 0, 0, 6, 1, False), (48, 12, 0, 4, 2, False),
                                     (48, 0, 20, 4, 3, False), (96, 24, 40, 5, 4, False), (200, 30, 90, 8, 5, False),
                                     (96, 24, 40, 0, 6, False), (60, 10, 16, 4, 7, True)]:
        out = lbfgs_direction_fixture(n, me, mi, m, seed, rd)
        np.savez_compressed(os.path.join(GOLD, "lbfgs_dir_n%d_me%d_mi%d_m%d_s%d.npz" % (n, me, mi, m, seed)), **out)
        print("lbfgs dir n=%d me=%d mi=%d m=%d |dz|=%.6g" % (n, me, mi, int(out["m"]), np.linalg.norm(out["dz_raw"])))


def run_example(k, x0, **kw):
    prob = example_problem(k)
    p = build(prob, x0, **kw)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        x, s, lda, fval, kkt = p.solve()
    d = pack_trace(p)
    d.update(problem=np.int64(k), x0=np.array(x0), x=x, s=s, lda=lda, fval=np.float64(fval),
             signal=np.int64(p.signal), nvar=np.int64(p.nvar), neq=np.int64(p.neq),
             nineq=np.int64(p.nineq), transcript=np.array(buf.getvalue()))
    for i, kk in enumerate(kkt):
        d["kkt%d" % (i + 1)] = np.atleast_1d(np.array(kk, dtype=np.float64))
    return d, p


def single_step(prob, x, mu=0.2, delta=0.0, s=None, lda=None, **kw):
    """One Newton step of the reference at a prescribed point (init rules of pyipm.py:1597-1625)."""
    p = build(prob, x, mu=mu, verbosity=-1, **kw)
    p.nvar = int(np.size(x))
    p.compile()
    x = np.array(x, dtype=np.float64)
    n, me, mi = p.nvar, p.neq, p.nineq
    if mi:
        s = p.init_slack(x) if s is None else np.array(s, dtype=np.float64)
        p.mu_host = mu
    else:
        s = np.array([], dtype=np.float64)
        p.mu_host = p.Ktol
        p.mu_dev.set_value(np.float64(p.mu_host))
    if me or mi:
        if lda is None:
            lda = p.init_lambda(x)
            li = lda[me:]
            li[li < 0.0] = p.Ktol
            lda[me:] = li
        else:
            lda = np.array(lda, dtype=np.float64)
    else:
        lda = np.array([], dtype=np.float64)
    p.delta = np.float64(delta)
    g = -p.grad(x, s, lda)
    H = p.hess(x, s, lda)
    Hc = p.reghess(np.array(H))
    dz_raw = p.sym_solve_cmp(Hc, g.reshape((g.size, 1))).reshape((g.size,))
    w = p.eigh(Hc)
    d = {"nvar": np.int64(n), "neq": np.int64(me), "nineq": np.int64(mi), "x": x, "s": s, "lda": lda,
         "mu": np.float64(p.mu_dev.get_value()), "mu_host": np.float64(p.mu_host),
         "delta_in": np.float64(delta), "delta_out": np.float64(p.delta),
         "g": g, "H": np.array(H), "Hc": np.array(Hc), "dz_raw": dz_raw,
         "dz": flipped(dz_raw, n, mi, me), "eig_Hc": w,
         "neg": np.int64(np.sum(w < -p.eps))}
    return d, p


def qp_step(n, me, mi, seed, keep_H):
    qp = make_qp(n, me, mi, seed)
    prob = qp_callables(qp)
    d, p = single_step(prob, qp["x"], mu=qp["mu"], s=qp["s"] if mi else None,
                       lda=qp["lam"] if (me or mi) else None)
    out = {k: d[k] for k in ("nvar", "neq", "nineq", "mu", "mu_host", "delta_in", "delta_out",
                             "g", "dz", "dz_raw", "neg")}
    out["seed"] = np.int64(seed)
    out["H_frob"] = np.float64(np.linalg.norm(d["H"]))
    out["H_trace"] = np.float64(np.trace(d["H"]))
    out["H_rowsum"] = d["H"].sum(axis=1)
    if keep_H:
        out["H"] = d["H"]
    return out


def main():
    os.makedirs(GOLD, exist_ok=True)
    x0s = unit_test_x0()

    # (ii) full-solve traces, seed-42 x0, the unit-test settings (Ftol=1e-8, unit_tests.py:50)
    for k in range(1, 11):
        d, p = run_example(k, x0s[k], Ftol=1.0e-8, verbosity=-1)
        np.savez_compressed(os.path.join(GOLD, "trace_p%02d.npz" % k), **d)
        print("p%-2d iters=%-3d signal=%2d x=%s" % (k, int(d["n_iter"]), int(d["signal"]), d["x"]))

    # README transcript shape for problem 7 (README.md:101-122): verbosity=1 banners
    d, p = run_example(7, x0s[7], Ftol=1.0e-8, verbosity=1)
    np.savez_compressed(os.path.join(GOLD, "transcript_p07.npz"), transcript=d["transcript"],
                        x=d["x"], n_iter=d["n_iter"], signal=d["signal"])

    # (i) problem 7 at a fixed point
    d, p = single_step(example_problem(7), [0.5, 0.3, 0.4], mu=0.2)
    np.savez_compressed(os.path.join(GOLD, "step_p07_fixed.npz"), **d)
    print("p7 fixed-point dz =", d["dz"])

    # (iii) synthetic QP KKT systems (generator of SURVEY.md section 8d)
    for (n, me, mi, seed) in [(24, 8, 16, 0), (40, 0, 12, 1), (40, 12, 0, 2), (64, 0, 0, 3),
                              (96, 32, 48, 4), (160, 40, 100, 5), (256, 64, 96, 6)]:
        out = qp_step(n, me, mi, seed, keep_H=(n + 2 * mi + me) <= 64)
        np.savez_compressed(os.path.join(GOLD, "qp_n%d_me%d_mi%d_s%d.npz" % (n, me, mi, seed)), **out)
        print("qp n=%d me=%d mi=%d N=%d delta_out=%g neg=%d" % (n, me, mi, n + 2 * mi + me,
                                                               out["delta_out"], out["neg"]))

    # (iv-a) nonconvex case that forces the delta x10 loop (pyipm.py:1399-1403)
    rng = np.random.default_rng(11)
    n, me, mi = 12, 3, 5
    Mx = rng.standard_normal((n, n))
    Qnc = 0.5 * (Mx + Mx.T) - 3.0 * np.eye(n)           # strongly indefinite Hessian
    Anc = rng.standard_normal((me, n))
    Gnc = rng.standard_normal((mi, n))
    cnc = rng.standard_normal(n)
    prob = {"nvar": n, "neq": me, "nineq": mi,
            "f": lambda x: 0.5 * x @ (Qnc @ x) + cnc @ x, "df": lambda x: Qnc @ x + cnc,
            "d2f": lambda x: Qnc,
            "ce": lambda x: Anc @ x - 0.1, "dce": lambda x: np.ascontiguousarray(Anc.T),
            "d2ce": lambda x, lda: np.zeros((n, n)),
            "ci": lambda x: Gnc @ x + 1.0, "dci": lambda x: np.ascontiguousarray(Gnc.T),
            "d2ci": lambda x, lda: np.zeros((n, n))}
    d, p = single_step(prob, np.zeros(n), mu=0.2)
    d.update(Q=Qnc, A=Anc, G=Gnc, c=cnc)
    np.savez_compressed(os.path.join(GOLD, "step_nonconvex_delta_loop.npz"), **d)
    print("nonconvex: delta_out=%g neg=%d (need %d)" % (d["delta_out"], d["neg"], me + mi))

    # (iv-b) rank-deficient Je that triggers the delta_c branch (pyipm.py:1383-1389)
    n, me, mi = 10, 4, 3
    Qr = np.eye(n) * 2.0
    Ar = rng.standard_normal((me, n))
    Ar[3] = Ar[0]                                        # duplicated equality row -> singular KKT
    Gr = rng.standard_normal((mi, n))
    cr = rng.standard_normal(n)
    prob = {"nvar": n, "neq": me, "nineq": mi,
            "f": lambda x: 0.5 * x @ (Qr @ x) + cr @ x, "df": lambda x: Qr @ x + cr,
            "d2f": lambda x: Qr,
            "ce": lambda x: Ar @ x - 0.1, "dce": lambda x: np.ascontiguousarray(Ar.T),
            "d2ce": lambda x, lda: np.zeros((n, n)),
            "ci": lambda x: Gr @ x + 1.0, "dci": lambda x: np.ascontiguousarray(Gr.T),
            "d2ci": lambda x, lda: np.zeros((n, n))}
    d, p = single_step(prob, np.zeros(n), mu=0.2)
    d.update(Q=Qr, A=Ar, G=Gr, c=cr)
    np.savez_compressed(os.path.join(GOLD, "step_rankdef_delta_c.npz"), **d)
    print("rank-deficient Je: delta_out=%g neg=%d Hc[le,le][0,0]=%g" %
          (d["delta_out"], d["neg"], d["Hc"][n + mi, n + mi]))


def pivot_step(name, n, me, mi, Q, seed, c_scale=1.0, blind=None):
    """One Newton step of the reference on  min 0.5 x'Qx + c'x  s.t.  A x = b,  G x - h >= 0  with a Hessian whose
    x-x tiles the device cannot pivot on their own (zero / zero-diagonal / zero rows).  Keeps the problem data, not
    the N x N matrices."""
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((me, n)) / np.sqrt(n)
    G = rng.standard_normal((mi, n)) / np.sqrt(n)
    if blind is not None:                      # constraint gradients orthogonal to `blind`: no constraint sees that direction
        A -= np.outer(A @ blind, blind)
        G -= np.outer(G @ blind, blind)
    c = c_scale * rng.standard_normal(n)
    x = 0.1 * rng.standard_normal(n)
    s = rng.uniform(0.5, 2.0, mi)
    lam = np.concatenate([rng.standard_normal(me), rng.uniform(0.5, 2.0, mi)])
    h = G @ x - s - 0.1 * rng.standard_normal(mi)
    b = A @ x - 0.1 * rng.standard_normal(me)
    Z = np.zeros((n, n))
    prob = {"nvar": n, "neq": me, "nineq": mi,
            "f": lambda x: 0.5 * x @ (Q @ x) + c @ x, "df": lambda x: Q @ x + c, "d2f": lambda x: Q,
            "ce": (lambda x: A @ x - b) if me else None, "dce": (lambda x: np.ascontiguousarray(A.T)) if me else None,
            "d2ce": (lambda x, lda: Z) if me else None,
            "ci": (lambda x: G @ x - h) if mi else None, "dci": (lambda x: np.ascontiguousarray(G.T)) if mi else None,
            "d2ci": (lambda x, lda: Z) if mi else None}
    d, p = single_step(prob, x, mu=0.2, s=s if mi else None, lda=lam if (me or mi) else None)
    out = {k: d[k] for k in ("nvar", "neq", "nineq", "mu", "mu_host", "delta_in", "delta_out", "g", "dz", "dz_raw", "neg",
                             "x", "s", "lda")}
    w = d["eig_Hc"]
    out.update(name=np.array(name), seed=np.int64(seed), Q=Q, A=A, G=G, c=c, b=b, h=h,
               rcond=np.float64(np.abs(w).min() / np.abs(w).max()), H_rowsum=d["H"].sum(axis=1),
               Hc_diag=np.diag(d["Hc"]).copy())
    return out


def main_pivot():
    """tests/golden/pivot_*.npz: systems whose x-x diagonal tiles are singular or indefinite on their own although the
    KKT matrix is fine -- the reference's LU pivots across the whole matrix (pyipm.py:18-20) and reghess leaves them
    unshifted (pyipm.py:1381) -- plus one where rcond fires although the inertia is right (pyipm.py:1379-1381)."""
    os.makedirs(GOLD, exist_ok=True)
    rng = np.random.default_rng(5)
    cases = []
    n = 128; cases.append(("lp", n, 0, 160, np.zeros((n, n)), 21))                         # d2L == 0, mi >= n
    n = 192; cases.append(("lp_eq", n, 48, 224, np.zeros((n, n)), 22))                     # LP with equalities
    n = 160
    M = 0.001 * rng.standard_normal((n, n)); Qz = M + M.T; np.fill_diagonal(Qz, 0.0)
    cases.append(("zerodiag", n, 16, 480, Qz, 23))                                         # zero diagonal, indefinite tiles
    n = 144
    M = rng.standard_normal((80, 80)); Ql = np.zeros((n, n)); Ql[:80, :80] = M @ M.T / 80 + np.eye(80)
    cases.append(("linear_vars", n, 12, 200, Ql, 24))                                      # 64 variables enter linearly
    for name, n, me, mi, Q, seed in cases:
        out = pivot_step(name, n, me, mi, Q, seed)
        np.savez_compressed(os.path.join(GOLD, "pivot_%s.npz" % name), **out)
        print("pivot %-12s N=%d delta_out=%g neg=%d (need %d) rcond=%.3g |dz|=%.6g" %
              (name, n + 2 * mi + me, out["delta_out"], out["neg"], me + mi, out["rcond"], np.linalg.norm(out["dz"])))
        assert out["delta_out"] == 0.0 and out["neg"] == me + mi
    # singular KKT matrix (pyipm.py:1379-1403): a variable with no curvature that no constraint touches -- an exactly
    # zero row, eigenvalue 0.  What the reference's test sees is the eigensolver's rounding of that zero (here
    # -3.2e-15: "one negative eigenvalue too many", rcond 6.9e-16 just above eps), so it shifts by delta without the
    # delta_c branch; either branch gives the same dz to ~delta_c = 8e-13.
    n, me, mi = 136, 8, 40
    M = rng.standard_normal((n, n)); Qr = M @ M.T / n + np.eye(n)
    k = 77
    Qr[k, :] = 0.0; Qr[:, k] = 0.0
    u = np.zeros(n); u[k] = 1.0
    out = pivot_step("rcond", n, me, mi, Qr, 25, blind=u)
    np.savez_compressed(os.path.join(GOLD, "pivot_singular.npz"), **out)
    print("pivot singular     delta_out=%g neg(after shift)=%d (need %d) Hc_diag[le]=%g" %
          (out["delta_out"], out["neg"], me + mi, out["Hc_diag"][n + mi]))
    assert out["delta_out"] > 0.0


def main_qp_trace():
    """tests/golden/qptrace_*.npz: the UNMODIFIED reference solving synthetic QPs end to end through its provider
    contract (pyipm.py:855-954: f, df, ce, ci, the Jacobians as compiled functions), every iterate at the Newton-step
    seam recorded -- what the device-resident QP loop (pyipm_amd/qp.py) has to retrace."""
    os.makedirs(GOLD, exist_ok=True)
    for (n, me, mi, seed, ktol) in [(40, 10, 24, 11, 1.0e-6), (64, 0, 48, 12, 1.0e-6), (48, 16, 0, 13, 1.0e-8)]:
        qp = make_qp(n, me, mi, seed)
        prob = qp_callables(qp)
        p = build(prob, np.zeros(n), Ktol=ktol, verbosity=-1)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            x, s, lda, fval, kkt = p.solve()
        d = {"n_iter": np.int64(len(p.trace)), "n": np.int64(n), "me": np.int64(me), "mi": np.int64(mi),
             "seed": np.int64(seed), "Ktol": np.float64(ktol)}
        for k in ("x", "s", "lda", "g", "dz_raw"):
            d["it_" + k] = np.stack([t[k] for t in p.trace])
        for k in ("mu", "mu_host", "delta_in", "delta_out"):
            d["it_" + k] = np.array([t[k] for t in p.trace])
        d.update(x=x, s=s, lda=lda, fval=np.float64(fval), signal=np.int64(p.signal))
        for i, kk in enumerate(kkt):
            d["kkt%d" % (i + 1)] = np.atleast_1d(np.array(kk, dtype=np.float64))
        np.savez_compressed(os.path.join(GOLD, "qptrace_n%d_me%d_mi%d_s%d.npz" % (n, me, mi, seed)), **d)
        print("qp trace n=%d me=%d mi=%d: %d Newton steps, signal %d, f = %.10g" % (n, me, mi, len(p.trace), p.signal, fval))


def lp_problem(n, mi_extra, seed):
    """min c'x  s.t.  G x - h >= 0: `mi_extra` random half-spaces around a strictly feasible point plus the box
    |x_i| <= 4 (so the LP is bounded) -- d2L == 0, mi = 2 n + mi_extra >= n."""
    rng = np.random.default_rng(seed)
    xf = rng.uniform(-1.0, 1.0, n)
    Gr = rng.standard_normal((mi_extra, n)) / np.sqrt(n)
    G = np.vstack([Gr, np.eye(n), -np.eye(n)])
    h = np.concatenate([Gr @ xf - rng.uniform(0.5, 1.5, mi_extra), -4.0 * np.ones(n), -4.0 * np.ones(n)])
    c = rng.standard_normal(n)
    return G, h, c, xf


def main_lp_trace():
    """tests/golden/lptrace_*.npz: the UNMODIFIED reference solving an LP end to end (d2L == 0: every x pivot of the device
    is a static pivot, at EVERY iterate, also the late ones where Sigma spans many decades), every iterate at the
    Newton-step seam recorded; and pivot_lp_late.npz: the single step of that trace with the widest Sigma (VERDICT r2
    item 4).  The reference's LU pivots over the whole matrix (pyipm.py:18-20) and never shifts here (pyipm.py:1381)."""
    os.makedirs(GOLD, exist_ok=True)
    n, mi_extra, seed = 128, 64, 31
    G, h, c, xf = lp_problem(n, mi_extra, seed)
    mi = G.shape[0]
    Z = np.zeros((n, n))
    GT = np.ascontiguousarray(G.T)
    prob = {"nvar": n, "neq": 0, "nineq": mi, "f": lambda x: c @ x, "df": lambda x: c.copy(), "d2f": lambda x: Z,
            "ce": None, "dce": None, "d2ce": None,
            "ci": lambda x: G @ x - h, "dci": lambda x: GT, "d2ci": lambda x, lda: Z}
    p = build(prob, xf, Ktol=1.0e-8, verbosity=-1)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        x, s, lda, fval, kkt = p.solve()
    d = {"n_iter": np.int64(len(p.trace)), "n": np.int64(n), "me": np.int64(0), "mi": np.int64(mi), "seed": np.int64(seed),
         "Ktol": np.float64(1.0e-8), "G": G, "h": h, "c": c, "x0": xf}
    for k in ("x", "s", "lda", "g", "dz_raw"):
        d["it_" + k] = np.stack([t[k] for t in p.trace])
    for k in ("mu", "mu_host", "delta_in", "delta_out"):
        d["it_" + k] = np.array([t[k] for t in p.trace])
    spread, neg, rcond = [], [], []
    for t in p.trace:
        sig = t["lda"] / (t["s"] + np.finfo(np.float64).eps)
        spread.append(sig.max() / sig.min())
        w = np.linalg.eigvalsh(t["Hc"])
        neg.append(int((w < 0).sum())); rcond.append(np.abs(w).min() / np.abs(w).max())
    d["it_sigma_spread"] = np.array(spread); d["it_neg"] = np.array(neg); d["it_rcond"] = np.array(rcond)
    d.update(x=x, s=s, lda=lda, fval=np.float64(fval), signal=np.int64(p.signal))
    for i, kk in enumerate(kkt):
        d["kkt%d" % (i + 1)] = np.atleast_1d(np.array(kk, dtype=np.float64))
    np.savez_compressed(os.path.join(GOLD, "lptrace_n%d_mi%d_s%d.npz" % (n, mi, seed)), **d)
    print("lp trace n=%d mi=%d: %d Newton steps, signal %d, f = %.10g, shifts %d, Sigma spread up to %.2e, rcond down to %.2e"
          % (n, mi, len(p.trace), p.signal, fval, int((d["it_delta_out"] > 0).sum()), max(spread), min(rcond)))
    # the late single step: widest Sigma among the iterates the reference left unshifted
    ok = [i for i in range(len(p.trace)) if d["it_delta_out"][i] == 0.0]
    i = max(ok, key=lambda j: spread[j])
    t = p.trace[i]
    out = {"name": np.array("lp_late"), "nvar": np.int64(n), "neq": np.int64(0), "nineq": np.int64(mi), "seed": np.int64(seed),
           "Q": Z, "A": np.zeros((0, n)), "G": G, "c": c, "b": np.zeros(0), "h": h, "x": t["x"], "s": t["s"], "lda": t["lda"],
           "mu": np.float64(t["mu"]), "mu_host": np.float64(t["mu_host"]), "delta_in": np.float64(t["delta_in"]),
           "delta_out": np.float64(t["delta_out"]), "g": t["g"], "dz_raw": t["dz_raw"], "dz": flipped(t["dz_raw"], n, mi, 0),
           "neg": np.int64(neg[i]), "rcond": np.float64(rcond[i]), "sigma_spread": np.float64(spread[i]), "iterate": np.int64(i),
           "H_rowsum": t["H"].sum(axis=1), "Hc_diag": np.diag(t["Hc"]).copy()}
    np.savez_compressed(os.path.join(GOLD, "pivot_lp_late.npz"), **out)
    print("pivot lp_late: iterate %d of %d, Sigma spread %.2e, rcond %.2e, neg %d (need %d), delta_in %g"
          % (i, len(p.trace), spread[i], rcond[i], neg[i], mi, t["delta_in"]))


if __name__ == "__main__":
    if "--lp-trace" in sys.argv[1:]:
        main_lp_trace()
    elif "--qp-trace" in sys.argv[1:]:
        main_qp_trace()
    elif "--pivot" in sys.argv[1:]:
        main_pivot()
    elif "--lbfgs" in sys.argv[1:]:
        main_lbfgs()
    else:
        main()
