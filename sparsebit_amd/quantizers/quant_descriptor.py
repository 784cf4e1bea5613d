"""(bit, scheme, target) -> integer range and axes.

Same public surface as sparsebit/quantization/quantizers/quant_descriptor.py:5-110:
qmin/qmax/qrange/bit/scheme/target/ch_axis/bs_axis/is_perchannel/is_symmetric,
set_bit, set_symmetric.
"""
import torch

from .. import plan as sbq_plan
from ..common import get_qscheme

_SYMMETRIC = (torch.per_channel_symmetric, torch.per_tensor_symmetric)
_PERCHANNEL = (torch.per_channel_symmetric, torch.per_channel_affine)
_LAYOUT_CH_AXIS = {"NCHW": 1, "NLC": 2}


class QuantDescriptor:
    def __init__(self, cfg):
        self._cfg = cfg
        self._target = cfg.TARGET[0]
        self._scheme = get_qscheme(cfg.QSCHEME)
        self._bit = cfg.QUANTIZER.BIT
        self.is_perchannel = self._scheme in _PERCHANNEL
        self.is_symmetric = self._scheme in _SYMMETRIC
        self._refresh_range()
        # weights: output channels lead; activations: given by the layout (quant_descriptor.py:36-58)
        layout = getattr(cfg.OBSERVER, "LAYOUT", None) if hasattr(cfg.OBSERVER, "LAYOUT") else None
        if layout is None:
            self._ch_axis, self._bs_axis = 0, None
        elif layout in _LAYOUT_CH_AXIS:
            self._ch_axis, self._bs_axis = _LAYOUT_CH_AXIS[layout], 0
        else:
            raise NotImplementedError(layout)

    @staticmethod
    def calc_qmin_qmax(bit, scheme):
        if scheme in _SYMMETRIC:  # int<b>:  [-2^(b-1), 2^(b-1) - 1]
            return -(2 ** (bit - 1)), 2 ** (bit - 1) - 1, "int{}".format(bit)
        return 0, 2 ** bit - 1, "uint{}".format(bit)  # uint<b>: [0, 2^b - 1]

    def _refresh_range(self):
        self._qmin, self._qmax, self._type = self.calc_qmin_qmax(self._bit, self._scheme)
        # the integer range is baked into launch plans (plan.QdqPlan keeps the version it was built for) and captured
        # graphs (the process-wide epoch): set_bit / set_symmetric on the descriptor itself invalidate both
        self.version = getattr(self, "version", 0) + 1
        sbq_plan.bump_epoch()

    def set_bit(self, bit):
        self._bit = bit
        self._refresh_range()

    def set_symmetric(self, is_symmetric: bool):
        self.is_symmetric = is_symmetric
        if self.is_perchannel:
            self._scheme = torch.per_channel_symmetric if is_symmetric else torch.per_channel_affine
        else:
            self._scheme = torch.per_tensor_symmetric if is_symmetric else torch.per_tensor_affine
        self._refresh_range()

    target = property(lambda self: self._target)
    scheme = property(lambda self: self._scheme)
    bit = property(lambda self: self._bit)
    qmin = property(lambda self: self._qmin)
    qmax = property(lambda self: self._qmax)
    qrange = property(lambda self: (self._qmin, self._qmax))
    ch_axis = property(lambda self: self._ch_axis)
    bs_axis = property(lambda self: self._bs_axis)

    def __repr__(self):
        return self._type + "\t qmin: {}  qmax: {}, qscheme: {}".format(self.qmin, self.qmax, self.scheme)
