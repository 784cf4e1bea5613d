"""Quantizer registry -- same contract as sparsebit/quantization/quantizers/__init__.py:1-23."""
QUANTIZERS_MAP = {}


def register_quantizer(quantizer):
    QUANTIZERS_MAP[quantizer.TYPE.lower()] = quantizer
    return quantizer


from .base import Quantizer  # noqa: E402
from . import uniform, lsq, lsq_plus, pact, dorefa  # noqa: E402,F401


def build_quantizer(cfg):
    # the reference asserts on the un-lowered name (quantizers/__init__.py:19-21); lower-casing
    # first accepts exactly a superset of it ("LSQ" and "lsq" both resolve)
    assert cfg.QUANTIZER.TYPE.lower() in QUANTIZERS_MAP, "no found an implement of {}".format(cfg.QUANTIZER.TYPE)
    quantizer = QUANTIZERS_MAP[cfg.QUANTIZER.TYPE.lower()](cfg)
    return quantizer
