# ``theano.compile.function_module`` is deliberately absent so pyipm.py:12-15 takes
# the AttributeError branch to ``compile.function.types.Function``.
from . import function  # noqa: F401
