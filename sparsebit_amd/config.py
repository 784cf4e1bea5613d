"""Minimal attribute-style config nodes.

The reference drives quantizers with yacs CfgNodes (sparsebit/quantization/
quant_config.py:6-48).  Only attribute access is needed on this path, so any
object exposing QSCHEME / QUANTIZER.{TYPE,BIT,DISABLE} / OBSERVER.{TYPE,LAYOUT,
PERCENTILE.ALPHA} / TARGET works -- including a real yacs node.  These helpers
build such nodes without yacs.
"""
from .common import QuantTarget


class Node(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def quantizer_config(qscheme, bit, quantizer="uniform", observer="MINMAX", target="weight", layout="NCHW",
                     alpha=1e-3, disable=False, ema_ratio=0.9, aciq="GAUS", pact_alpha=10):
    """One side (W or A) of a qconfig, already carrying TARGET like QuantOpr.build_quantizer
    sets it (sparsebit/quantization/modules/base.py:36-45)."""
    obs = Node(TYPE=observer, PERCENTILE=Node(ALPHA=alpha), ACIQ=Node(DISTRIBUTION=aciq))
    if target != "weight":
        obs["LAYOUT"] = layout  # activations only: QuantDescriptor keys ch_axis off its presence
        obs["MOVING_AVERAGE"] = Node(EMA_RATIO=ema_ratio)
    return Node(
        QSCHEME=qscheme,
        QUANTIZER=Node(TYPE=quantizer, BIT=bit, DISABLE=disable, PACT=Node(ALPHA_VALUE=pact_alpha)),
        OBSERVER=obs,
        TARGET=(QuantTarget.WEIGHT,) if target == "weight" else (QuantTarget.FEATURE,),
    )


def sparser_config(ratio, type_="unstructed", strategy="l1norm"):
    """sparsebit/sparse/sparse_config.py:5-15 ("unstructed" is the reference's spelling)."""
    return Node(SPARSER=Node(TYPE=type_, STRATEGY=strategy, RATIO=ratio))
