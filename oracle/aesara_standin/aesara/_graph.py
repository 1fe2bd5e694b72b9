"""Minimal lazy expression graph evaluated with NumPy.  TEST INFRASTRUCTURE ONLY.

This is NOT Aesara and NOT part of the product.  It exists so that the UNMODIFIED
reference file /root/reference/pyipm.py can be imported and driven in the build
container (where the real ``aesara`` package is absent) in order to generate
golden traces for ``tests/golden`` (see ``oracle/make_golden.py``).  Only the
small API surface pyipm.py touches is provided; there is no autodiff
(``T.grad`` / ``gradient.hessian`` / ``gradient.jacobian`` raise), so every
derivative must be handed to the reference as an expression or as a
"precompiled" Function — both are input states the reference supports
(pyipm.py:216-221, 426-440).

Terminal numerics: ``slinalg.solve`` -> scipy.linalg.solve(assume_a=...),
``slinalg.eigvalsh`` -> scipy.linalg.eigvalsh(a, b, lower=True),
``nlinalg.pinv`` -> numpy.linalg.pinv — the SciPy/NumPy routines Aesara's own
Ops forward to.
"""
import numpy as np


class Expr(object):
    __array_priority__ = 1000.0   # make ndarray <op> Expr defer to Expr.__r<op>__

    def __init__(self, fn, args=(), name=None):
        self._fn = fn
        self._args = tuple(args)
        self.name = name

    # -- evaluation -----------------------------------------------------------
    def eval_in(self, env, cache):
        key = id(self)
        if key in cache:
            return cache[key]
        if self in env:
            val = env[self]
        else:
            vals = [a.eval_in(env, cache) if isinstance(a, Expr) else a for a in self._args]
            val = self._fn(*vals)
        cache[key] = val
        return val

    def __hash__(self):
        return id(self)

    def __eq__(self, other):
        return self is other

    # -- arithmetic -----------------------------------------------------------
    def __add__(self, o): return Expr(lambda a, b: a + b, (self, o))
    def __radd__(self, o): return Expr(lambda a, b: b + a, (self, o))
    def __sub__(self, o): return Expr(lambda a, b: a - b, (self, o))
    def __rsub__(self, o): return Expr(lambda a, b: b - a, (self, o))
    def __mul__(self, o): return Expr(lambda a, b: a * b, (self, o))
    def __rmul__(self, o): return Expr(lambda a, b: b * a, (self, o))
    def __truediv__(self, o): return Expr(lambda a, b: a / b, (self, o))
    def __rtruediv__(self, o): return Expr(lambda a, b: b / a, (self, o))
    def __pow__(self, o): return Expr(lambda a, b: a ** b, (self, o))
    def __neg__(self): return Expr(lambda a: -a, (self,))

    def __getitem__(self, idx):
        # slice bounds may be lazy (``WT_g[:m_lbfgs]``, pyipm.py:1169)
        parts = idx if isinstance(idx, tuple) else (idx,)
        lazy = []
        for q in parts:
            if isinstance(q, slice):
                lazy += [v for v in (q.start, q.stop, q.step) if isinstance(v, Expr)]
            elif isinstance(q, Expr):
                lazy.append(q)

        def f(a, *vals):
            if not vals:
                return a[idx]
            it = iter(vals)
            pick = lambda v: int(next(it)) if isinstance(v, Expr) else v      # noqa: E731
            out = tuple(slice(pick(q.start), pick(q.stop), pick(q.step)) if isinstance(q, slice) else pick(q)
                        for q in parts)
            return a[out if isinstance(idx, tuple) else out[0]]
        sub = Expr(f, (self,) + tuple(lazy))
        sub._index = idx
        return sub

    @property
    def shape(self):
        return _Shape(self)

    @property
    def T(self):
        return Expr(lambda a: a.T, (self,))

    @property
    def size(self):
        return Expr(lambda a: int(np.size(a)), (self,))

    def reshape(self, shp):
        # shape entries may themselves be lazy (e.g. ``Adiag.reshape((Adiag.size,))``, pyipm.py:1103)
        dims = tuple(shp) if isinstance(shp, (tuple, list)) else (shp,)
        lazy = [d for d in dims if isinstance(d, Expr)]

        def f(a, *vals):
            it = iter(vals)
            return np.reshape(a, tuple(int(next(it)) if isinstance(d, Expr) else d for d in dims))
        return Expr(f, (self,) + tuple(lazy))

    def ravel(self):
        return Expr(lambda a: np.ravel(a), (self,))


class _Shape(object):
    def __init__(self, e):
        self._e = e

    def __getitem__(self, i):
        return Expr(lambda a: np.shape(a)[i], (self._e,))


class Variable(Expr):
    def __init__(self, name=None, ndim=1):
        Expr.__init__(self, None, (), name)
        self.ndim = ndim

    def eval_in(self, env, cache):
        if self in env:
            return env[self]
        raise KeyError("unbound stand-in variable %r" % (self.name,))


class SharedVariable(Expr):
    def __init__(self, value, name=None):
        Expr.__init__(self, None, (), name)
        self._value = value

    def get_value(self):
        return self._value

    def set_value(self, v):
        self._value = v

    def eval_in(self, env, cache):
        return self._value


class Function(object):
    """Stand-in for aesara.compile.function.types.Function (pyipm.py:12-15)."""

    def __init__(self, inputs, outputs, on_unused_input=None):
        self.inputs = list(inputs)
        self.outputs = outputs

    def __call__(self, *vals):
        if len(vals) != len(self.inputs):
            raise TypeError("expected %d inputs, got %d" % (len(self.inputs), len(vals)))
        env = {}
        for var, val in zip(self.inputs, vals):
            env[var] = np.asarray(val)
        cache = {}
        out = self.outputs
        if isinstance(out, Expr):
            return np.array(out.eval_in(env, cache))
        return np.asarray(out)


def function(inputs=None, outputs=None, on_unused_input=None, **kw):
    return Function(inputs, outputs, on_unused_input)


def shared(value, name=None, **kw):
    return SharedVariable(value, name)


def wrap(fn, *args):
    """Build an expression node from a NumPy callable (how tests hand derivative
    'expressions' to the reference without autodiff)."""
    return Expr(fn, args)
