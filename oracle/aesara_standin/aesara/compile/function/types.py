from ..._graph import Function  # noqa: F401
