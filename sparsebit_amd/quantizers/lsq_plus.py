"""LSQ+ quantizer (mirrors sparsebit/quantization/quantizers/lsq_plus.py:13-96).

Weights (per-channel symmetric): scale from the per-channel mean and unbiased std, both
out of ONE moments pass (sum x, sum x^2 in fp64).  Activations (per-tensor affine): scale and
zero_point from the configured observer, both learnable."""
import math

import torch
import torch.nn as nn

from . import Quantizer as BaseQuantizer
from . import register_quantizer
from .. import dist as sbq_dist
from .. import ops
from .lsq import gs_scaling
from .quant_tensor import STE


@register_quantizer
class Quantizer(BaseQuantizer):
    TYPE = "LSQ+"

    def __init__(self, config):
        super(Quantizer, self).__init__(config)
        self.init_params = False

    def calc_qparams(self):
        if self.fake_fused:
            return self.scale, self.zero_point
        if not self.init_params:
            if self.is_perchannel:
                assert self.is_symmetric, "LSQ+ only support per-channel-symmetric quant for weight"
                shards = self.observer._shards()
                ch_axis = self.qdesc.ch_axis
                C = shards[0].shape[ch_axis]
                s1 = torch.zeros(C, dtype=torch.float64, device=shards[0].device)
                s2 = torch.zeros_like(s1)
                n = 0
                for x in shards:
                    ops.channel_moments(x, ch_axis, True, s1, s2)
                    n += x.numel() // C
                sbq_dist.allreduce_sum_(s1)
                sbq_dist.allreduce_sum_(s2)
                n = sbq_dist.allreduce_count(n)
                self.observer.data_cache.reset()
                mean64 = s1 / n
                var64 = torch.clamp(s2 - s1 * mean64, min=0.0) / max(n - 1, 1)  # torch.std: unbiased
                mean, std = mean64.float(), var64.sqrt().float()
                scale = 2 * torch.maximum((mean - 3 * std).abs(), (mean + 3 * std).abs()) / (
                    self.qdesc.qmax - self.qdesc.qmin
                )
                self.scale = nn.Parameter(self._broadcast_qparams(scale.to(self.device)))
                self.zero_point = self._broadcast_qparams(torch.zeros_like(self.scale))
            else:
                assert not self.is_symmetric, "LSQ+ only support per-tensor-affine quant for activation"
                scale, zero_point = self.observer.calc_qparams()
                self.scale = nn.Parameter(self._broadcast_qparams(scale.to(self.device)))
                zero_point = zero_point.clamp(self.qdesc.qmin, self.qdesc.qmax)
                self.zero_point = nn.Parameter(self._broadcast_qparams(zero_point.to(self.device)))
            self.init_params = True
        return self.scale, self.zero_point

    def _qparams_preprocess(self, x):
        if self.export_onnx:
            return (
                torch.tensor(self.scale.abs().detach().cpu().numpy(), device=self.device),
                torch.tensor(
                    torch.clamp(self.zero_point, self.qdesc.qmin, self.qdesc.qmax).detach().cpu().numpy(),
                    device=self.device,
                ),
            )
        scale = self.scale.abs()
        zero_point = torch.clamp(self.zero_point, self.qdesc.qmin, self.qdesc.qmax)
        return scale, zero_point

    def _forward(self, x, scale, zero_point):
        if self.is_perchannel:
            num_perchannel = x.numel() / x.shape[self.qdesc.ch_axis]
            gs_ratio = 1.0 / math.sqrt(num_perchannel * self.qdesc.qmax)
        else:
            gs_ratio = 1.0 / math.sqrt(x.numel() * self.qdesc.qmax)
        scale = gs_scaling.apply(scale, gs_ratio)
        if zero_point.requires_grad:
            zero_point = gs_scaling.apply(zero_point, gs_ratio)
        return STE.apply(x, scale, zero_point, self.qdesc, self.backend)
