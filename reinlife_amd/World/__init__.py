from .environment import Environment  # noqa: F401
