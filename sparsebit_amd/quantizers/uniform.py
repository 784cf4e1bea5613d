"""Uniform affine/symmetric quantizer: the straight-through fake-quant with observer-provided
scale / zero_point (interface of sparsebit/quantization/quantizers/uniform.py:7-16), plus the
real integer view of a tensor, which the fake-quant kernel can emit in the same pass."""
import torch

from . import Quantizer as BaseQuantizer
from . import register_quantizer
from .. import ops
from .quant_tensor import ste_fake_quant


@register_quantizer
class Quantizer(BaseQuantizer):
    TYPE = "uniform"

    def _forward(self, x_f, scale, zero_point):
        return ste_fake_quant(x_f, scale, zero_point, self.qdesc, self.backend, self._out_dtype(x_f))

    @torch.no_grad()
    def quantize_to_int(self, x, dtype=None):
        """(dequantized, integer levels) from one kernel launch; integer dtype defaults to
        int8 / uint8 when the range fits, else int32."""
        lo, hi = self.qdesc.qrange
        if dtype is None:
            dtype = torch.int32 if hi - lo > 255 else (torch.int8 if lo < 0 else torch.uint8)
        scale, zero_point = self._qparams_preprocess(x)
        return ops.fake_quant(x, scale, zero_point, lo, hi, self.qdesc.ch_axis, return_q=dtype)
