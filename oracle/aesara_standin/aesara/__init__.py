"""Test-only stand-in for the absent third-party ``aesara`` package — see _graph.py."""
from ._graph import Expr, Variable, SharedVariable, Function, function, shared, wrap  # noqa: F401
from . import tensor  # noqa: F401
from . import compile  # noqa: F401
from . import ifelse  # noqa: F401
from . import gradient  # noqa: F401

__version__ = "0.0-standin"
