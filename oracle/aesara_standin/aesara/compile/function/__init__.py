from . import types  # noqa: F401
