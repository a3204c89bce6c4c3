"""reinlife_amd -- MI355X-native implementation of ReinLife's per-tick hot path (world tick + policy inference).

Drop-in surface (same names/keywords as the reference, ReinLife/__init__.py:1-5):
    from reinlife_amd import trainer, tester, Environment, Models
"""
from . import Models  # noqa: F401
from . import _lib  # noqa: F401
from .Helpers.tester import tester  # noqa: F401
from .Helpers.trainer import trainer  # noqa: F401
from .World.environment import Environment  # noqa: F401

__all__ = ["Models", "trainer", "tester", "Environment"]
