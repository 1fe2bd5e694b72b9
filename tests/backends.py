"""Test-only Newton backends (checkers, never shipped in pyipm_amd/)."""
import numpy as np

from oracle import newton_oracle as orc


class OracleBackend(object):
    """Implements the backend interface of pyipm_amd.ipm with the CPU oracle (reference path:
    eigen-inertia reghess + LU).  Lets the CPU suite exercise the HOST loop / line search."""

    def __init__(self, n, me, mi):
        self.n, self.me, self.mi = n, me, mi
        self.calls = []

    def direction(self, d2L, Je, Ji, df, ce, ci, s, lda, mu, delta, mu_host, eta, beta, reg_coef, delta0, eps):
        st = {}
        dz, delta, Hc, g = orc.newton_step(d2L, Je, Ji, df, ce, ci, s, lda, mu, self.n, self.me, self.mi,
                                           delta=delta, mu_host=mu_host, eta=eta, beta=beta, eps=eps, stats=st)
        self.calls.append({"Hc": Hc, "g": g, "dz": dz, "delta": delta})
        return dz, delta, st
