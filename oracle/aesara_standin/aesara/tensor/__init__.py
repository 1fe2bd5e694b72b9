import numpy as np
from .._graph import Expr, Variable
from . import slinalg, nlinalg, basic  # noqa: F401
from .basic import diag  # noqa: F401


def vector(name=None, **kw):
    return Variable(name, 1)


def matrix(name=None, **kw):
    return Variable(name, 2)


def _e(fn, *args):
    return Expr(fn, args)


def sum(x, axis=None): return _e(lambda a: np.sum(a, axis=axis), x)          # noqa: A001
def log(x): return _e(np.log, x)
def abs_(x): return _e(np.abs, x)
def dot(a, b): return _e(np.dot, a, b)
def eye(n, *a): return _e(lambda k: np.eye(int(k)), n)
def zeros(shape, **kw): return _e(lambda: np.zeros(shape))
def ones(shape, **kw): return _e(lambda: np.ones(shape))
def triu(x, k=0): return _e(lambda a: np.triu(a, k), x)
def diagonal(x): return _e(np.diagonal, x)
def max(x, axis=None): return _e(lambda a: np.max(a, axis=axis), x)          # noqa: A001
def concatenate(xs, axis=0): return Expr(lambda *a: np.concatenate(a, axis=axis), tuple(xs))


class _Sub(object):
    """x[idx] remembered so set_subtensor/inc_subtensor know parent and index."""


def _split(sub):
    # ``sub`` was produced by Expr.__getitem__: recover (parent, idx) from its closure
    parent = sub._args[0]
    idx = sub._fn.__closure__[0].cell_contents
    return parent, idx


def set_subtensor(sub, val):
    parent, idx = _split(sub)

    def f(p, v):
        out = np.array(p, dtype=np.float64, copy=True)
        out[idx] = v
        return out
    return Expr(f, (parent, val))


def inc_subtensor(sub, val):
    parent, idx = _split(sub)

    def f(p, v):
        out = np.array(p, dtype=np.float64, copy=True)
        out[idx] = out[idx] + v
        return out
    return Expr(f, (parent, val))


def grad(*a, **k):
    raise NotImplementedError("aesara stand-in has no autodiff; pass derivatives explicitly")
