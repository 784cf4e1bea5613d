"""Quantizer plug-in point: `register_quantizer`, `QUANTIZERS_MAP`, `build_quantizer(cfg)` --
the names and behaviour of sparsebit/quantization/quantizers/__init__.py:1-23."""
from ..registry import Registry

QUANTIZERS_MAP = Registry("quantizer", "TYPE")
register_quantizer = QUANTIZERS_MAP.register

from .base import Quantizer  # noqa: E402
from . import uniform, lsq, lsq_plus, pact, dorefa  # noqa: E402,F401


def build_quantizer(cfg):
    """cfg.QUANTIZER.TYPE picks the class (case-insensitively: a superset of the reference,
    which asserts on the un-lowered name)."""
    return QUANTIZERS_MAP.resolve(cfg.QUANTIZER.TYPE)(cfg)
