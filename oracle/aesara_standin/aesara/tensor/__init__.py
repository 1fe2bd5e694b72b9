import numpy as np
from .._graph import Expr, Variable
from . import slinalg, nlinalg, basic  # noqa: F401
from .basic import diag  # noqa: F401


def vector(name=None, **kw):
    return Variable(name, 1)


def scalar(name=None, **kw):
    return Variable(name, 0)


def matrix(name=None, **kw):
    return Variable(name, 2)


def _e(fn, *args):
    return Expr(fn, args)


def sum(x, axis=None): return _e(lambda a: np.sum(a, axis=axis), x)          # noqa: A001
def log(x): return _e(np.log, x)
def abs_(x): return _e(np.abs, x)
def dot(a, b): return _e(np.dot, a, b)
def eye(n, *a): return _e(lambda k: np.eye(int(k)), n)
def _shaped(ctor, shape):
    # shape entries may be lazy (``T.zeros((self.nineq, 2 * m_lbfgs))``, pyipm.py:1082)
    dims = tuple(shape) if isinstance(shape, (tuple, list)) else (shape,)
    lazy = [d for d in dims if isinstance(d, Expr)]

    def f(*vals):
        it = iter(vals)
        return ctor(tuple(int(next(it)) if isinstance(d, Expr) else d for d in dims))
    return Expr(f, tuple(lazy))


def zeros(shape, **kw): return _shaped(np.zeros, shape)
def ones(shape, **kw): return _shaped(np.ones, shape)
def min(x, axis=None): return _e(lambda a: np.min(a, axis=axis), x)          # noqa: A001
def gt(a, b): return _e(lambda u, v: u > v, a, b)
def le(a, b): return _e(lambda u, v: u <= v, a, b)
def triu(x, k=0): return _e(lambda a: np.triu(a, k), x)
def diagonal(x): return _e(np.diagonal, x)
def max(x, axis=None): return _e(lambda a: np.max(a, axis=axis), x)          # noqa: A001
def concatenate(xs, axis=0): return Expr(lambda *a: np.concatenate(a, axis=axis), tuple(xs))


class _Sub(object):
    """x[idx] remembered so set_subtensor/inc_subtensor know parent and index."""


def _split(sub):
    # ``sub`` was produced by Expr.__getitem__: recover (parent, idx) from its closure
    parent = sub._args[0]
    idx = sub._index
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
