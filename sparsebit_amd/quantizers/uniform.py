"""uniform quantizer (mirrors sparsebit/quantization/quantizers/uniform.py:7-16)."""
from . import Quantizer as BaseQuantizer
from . import register_quantizer
from .quant_tensor import STE


@register_quantizer
class Quantizer(BaseQuantizer):
    TYPE = "uniform"

    def __init__(self, config):
        super(Quantizer, self).__init__(config)

    def _forward(self, x_f, scale, zero_point):
        return STE.apply(x_f, scale, zero_point, self.qdesc, self.backend)
