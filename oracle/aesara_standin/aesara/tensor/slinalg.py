import numpy as np
import scipy.linalg
from .._graph import Expr


def solve(a, b, assume_a='gen', **kw):
    """What aesara.tensor.slinalg.Solve.perform forwards to."""
    return Expr(lambda A, B: scipy.linalg.solve(A, B, assume_a=assume_a), (a, b))


def eigvalsh(a, b, lower=True):
    """What aesara.tensor.slinalg.Eigvalsh.perform forwards to (generalised form)."""
    return Expr(lambda A, B: scipy.linalg.eigvalsh(A, B, lower=lower), (a, b))
