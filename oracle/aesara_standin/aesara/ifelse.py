from ._graph import Expr


def ifelse(cond, a, b):
    return Expr(lambda c, x, y: x if c else y, (cond, a, b))
