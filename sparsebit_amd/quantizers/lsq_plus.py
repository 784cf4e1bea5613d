"""LSQ+ (interface and initialisation of sparsebit/quantization/quantizers/lsq_plus.py:13-96).

It is LSQ with a different initialisation -- weights (per-channel symmetric): the scale
covers mean +- 3 std of each output channel; activations (per-tensor affine): scale AND
zero_point start from the configured observer and are both learned -- so everything but
`calc_qparams` is inherited from the LSQ quantizer (abs / clamp preprocessing, gradient
scaling of scale and, when it is a Parameter, of the zero point, shared STE kernel).
Mean and unbiased std of every channel come out of ONE fp64 moments pass over the weight.
"""
import torch
import torch.nn as nn

from . import register_quantizer
from .. import dist as sbq_dist
from .. import ops
from .lsq import Quantizer as LSQQuantizer


@register_quantizer
class Quantizer(LSQQuantizer):
    TYPE = "LSQ+"

    def _channel_mean_std(self):
        shards = self.observer._shards()
        axis = self.qdesc.ch_axis
        C = shards[0].shape[axis]
        s1 = torch.zeros(C, dtype=torch.float64, device=shards[0].device)
        s2 = torch.zeros_like(s1)
        count = 0
        for x in shards:
            ops.channel_moments(x, axis, True, s1, s2)
            count += x.numel() // C
        sbq_dist.allreduce_sum_(s1)
        sbq_dist.allreduce_sum_(s2)
        count = sbq_dist.allreduce_count(count)
        self.observer.data_cache.reset()
        mean = s1 / count
        var = torch.clamp(s2 - s1 * mean, min=0.0) / max(count - 1, 1)  # torch.std is unbiased
        return mean.float(), var.sqrt().float()

    def calc_qparams(self):
        if self.fake_fused or self.init_params:
            return self.scale, self.zero_point
        lo, hi = self.qdesc.qrange
        if self.is_perchannel:
            assert self.is_symmetric, "LSQ+ only support per-channel-symmetric quant for weight"
            mean, std = self._channel_mean_std()
            reach = torch.maximum((mean - 3 * std).abs(), (mean + 3 * std).abs())
            self.scale = nn.Parameter(self._broadcast_qparams((2 * reach / (hi - lo)).to(self.device)))
            self.zero_point = self._broadcast_qparams(torch.zeros_like(self.scale))
        else:
            assert not self.is_symmetric, "LSQ+ only support per-tensor-affine quant for activation"
            scale, zero_point = self.observer.calc_qparams()
            self.scale = nn.Parameter(self._broadcast_qparams(scale.to(self.device)))
            self.zero_point = nn.Parameter(self._broadcast_qparams(zero_point.clamp(lo, hi).to(self.device)))
        self.init_params = True
        return self.scale, self.zero_point
