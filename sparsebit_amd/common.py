"""Enums and scheme parsing -- same names and meaning as sparsebit/quantization/common.py."""
from enum import Enum

import torch


class Granularity(Enum):
    LAYERWISE = 0
    CHANNELWISE = 1


class QuantTarget(Enum):
    WEIGHT = 0
    FEATURE = 1


class Backend(Enum):
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
