"""Enums and scheme parsing -- same names and meaning as sparsebit/quantization/common.py.

The reference hands quantizers ITS OWN enum members: `QuantOpr.build_quantizer` sets
`cfg.TARGET = (QuantTarget.WEIGHT,)` and calls `set_backend(get_backend(...))` with the classes
of sparsebit/quantization/common.py:5-35 (modules/base.py:36-45).  Two `Enum` classes never
compare equal by default, so the enums here compare and hash BY NAME: a member equals any enum
member of a same-named class with the same member name.  `Enum.__hash__` is already
`hash(name)`, so dict lookups keyed by these members (`fake_quant_factory[backend]`) find the
entry for a reference member too; `==` falls back to our reflected `__eq__` because the plain
Enum on the other side answers NotImplemented.
"""
from enum import Enum

import torch


class _ByName(Enum):
    def __eq__(self, other):
        if self is other:
            return True
        if isinstance(other, Enum) and type(other).__name__ == type(self).__name__:
            return other.name == self.name
        return NotImplemented

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else not r

    def __hash__(self):
        return hash(self._name_)


class Granularity(_ByName):
    LAYERWISE = 0
    CHANNELWISE = 1


class QuantTarget(_ByName):
    WEIGHT = 0
    FEATURE = 1


class Backend(_ByName):
    VIRTUAL = 0
    ONNXRUNTIME = 1
    TENSORRT = 2


_BACKENDS = {"virtual": Backend.VIRTUAL, "onnxruntime": Backend.ONNXRUNTIME, "tensorrt": Backend.TENSORRT}
_QSCHEMES = {
    "per-tensor-symmetric": torch.per_tensor_symmetric,
    "per-tensor-affine": torch.per_tensor_affine,
    "per-channel-symmetric": torch.per_channel_symmetric,
    "per-channel-affine": torch.per_channel_affine,
}


def get_backend(backend):
    try:
        return _BACKENDS[backend]
    except KeyError:
        raise TypeError("only support backend in {}, not {}".format(sorted(_BACKENDS), backend))


def get_qscheme(qscheme):
    try:
        return _QSCHEMES[qscheme]
    except KeyError:
        raise TypeError(
            "only support a qscheme equals to per-[tensor/channel]-[affine/symmetric] , not {}".format(qscheme)
        )
