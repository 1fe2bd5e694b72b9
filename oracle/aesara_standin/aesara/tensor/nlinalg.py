import numpy as np
from .._graph import Expr


def pinv(x):
    return Expr(lambda a: np.linalg.pinv(a), (x,))


def eigh(x):
    return Expr(lambda a: np.linalg.eigh(a), (x,))
