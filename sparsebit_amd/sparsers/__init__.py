"""Sparser plug-in point: `register_sparser`, `SPARSERS_MAP`, `build_sparser(config, opr)` --
the names and behaviour of sparsebit/sparse/sparsers/__init__.py:1-18."""
from ..registry import Registry

SPARSERS_MAP = Registry("sparser", "STRATEGY")
register_sparser = SPARSERS_MAP.register

from .base import Sparser  # noqa: E402
from . import l1norm  # noqa: E402,F401


def build_sparser(config, opr=None):
    return SPARSERS_MAP.resolve(config.SPARSER.STRATEGY)(config, opr=opr)
