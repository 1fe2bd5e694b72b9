from ._graph import Expr


class _IfElse(Expr):
    """Lazy like Aesara's IfElse: only the taken branch is evaluated (pyipm.py:1097, 1148, 1175 rely on it:
    with an empty L-BFGS memory the other branch would solve 0 x 0 systems)."""

    def __init__(self, cond, a, b):
        Expr.__init__(self, None, (cond, a, b))

    def eval_in(self, env, cache):
        key = id(self)
        if key in cache:
            return cache[key]
        cond, a, b = self._args
        c = cond.eval_in(env, cache) if isinstance(cond, Expr) else cond
        pick = a if c else b
        val = pick.eval_in(env, cache) if isinstance(pick, Expr) else pick
        cache[key] = val
        return val


def ifelse(cond, a, b):
    return _IfElse(cond, a, b)
