"""Sparser registry -- same contract as sparsebit/sparse/sparsers/__init__.py:1-18."""
SPARSERS_MAP = {}


def register_sparser(sparser):
    SPARSERS_MAP[sparser.STRATEGY.lower()] = sparser
    return sparser


from .base import Sparser  # noqa: E402
from . import l1norm  # noqa: E402,F401


def build_sparser(config, opr=None):
    assert config.SPARSER.STRATEGY.lower() in SPARSERS_MAP, "no found an implement of {}".format(
        config.SPARSER.STRATEGY
    )
    return SPARSERS_MAP[config.SPARSER.STRATEGY.lower()](config, opr=opr)
