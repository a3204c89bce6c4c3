"""reinlife_amd -- MI355X-native implementation of ReinLife's per-tick hot path (world tick + policy inference).

Drop-in surface (same names/keywords as the reference, ReinLife/__init__.py:1-5):
    from reinlife_amd import trainer, tester, Environment, Models
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
