"""DoReFa weight quantizer (behaviour of sparsebit/quantization/quantizers/dorefa.py:8-27):
squash with tanh, rescale to [-1, 1] by the largest magnitude, fake-quantize.  The largest
magnitude is one sbq_channel_stats reduction; the observer is fed the squashed tensor so that
calibration sees what the forward pass will quantize."""
import torch

from . import Quantizer as BaseQuantizer
from . import register_quantizer
from .. import ops
from .quant_tensor import STE


def _squash(x):
    """tanh(x) / max|tanh(x)|; the divisor is a constant of the graph (detached)."""
    t = x.tanh()
    lo, hi, _ = ops.channel_stats(t.detach(), 0, False)
    return t / torch.maximum(hi, -lo).reshape(())


@register_quantizer
class Quantizer(BaseQuantizer):
    TYPE = "DoReFa"

    def update_observer(self, x):
        self.dims = x.dim()
        x = x.detach()
        self.observer.data_cache.update(_squash(x if x.is_cuda else x.to(self.device)))

    def _forward(self, x, scale, zero_point):
        # like the reference, the stored parameters are used, not the preprocessed ones
        return STE.apply(_squash(x), self.scale, self.zero_point, self.qdesc, self.backend, self._out_dtype(x))
