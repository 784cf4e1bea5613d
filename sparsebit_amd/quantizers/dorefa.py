"""DoReFa quantizer (mirrors sparsebit/quantization/quantizers/dorefa.py:8-27): tanh, scale
to [-1, 1] by the tensor's largest magnitude (one sbq_channel_stats pass), shared STE kernel."""
import torch

from . import Quantizer as BaseQuantizer
from . import register_quantizer
from .. import ops
from .quant_tensor import STE


def _absmax(t):
    mn, mx, _ = ops.channel_stats(t.detach(), 0, False)
    return torch.maximum(mx, -mn).reshape(())


@register_quantizer
class Quantizer(BaseQuantizer):
    TYPE = "DoReFa"

    def __init__(self, config):
        super(Quantizer, self).__init__(config)

    def _forward(self, x, scale, zero_point):
        x_tanhed = x.tanh()
        x_normed = x_tanhed / _absmax(x_tanhed)  # norm to [-1, +1]
        scale, zero_point = self.scale, self.zero_point
        return STE.apply(x_normed, scale, zero_point, self.qdesc, self.backend)

    def update_observer(self, x):
        self.dims = len(x.shape)
        if not x.is_cuda:
            x = x.to(self.device)
        x_tanhed = x.detach().tanh()
        self.observer.data_cache.update(x_tanhed / _absmax(x_tanhed))
