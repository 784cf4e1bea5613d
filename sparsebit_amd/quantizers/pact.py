"""PACT quantizer (mirrors sparsebit/quantization/quantizers/pact.py:12-45): a learnable
clipping bound alpha in front of the shared STE kernel."""
import torch
import torch.nn as nn

from . import Quantizer as BaseQuantizer
from . import register_quantizer
from .. import ops
from ..common import QuantTarget
from .quant_tensor import STE


class _QParamsFromBounds(torch.autograd.Function):
    """scale / zero_point from (lower, upper) through the exact device kernel, with the
    gradient the reference gets from running observers/base.py:63-79 as autograd-tracked
    torch ops: only `lower` carries grad (pact.py:33-41 detaches alpha), scale depends on it
    through min(lower, 0); torch.maximum splits the gradient evenly at a tie, which is the
    symmetric case's permanent state (-lower == alpha)."""

    @staticmethod
    def forward(ctx, lower, upper, qmin, qmax, symmetric):
        scale, zero_point = ops.qparams_from_minmax(lower.detach(), upper, qmin, qmax, symmetric)
        ctx.save_for_backward(lower.detach(), upper, scale)
        ctx.qrange = float(qmax - qmin)
        ctx.symmetric = symmetric
        ctx.mark_non_differentiable(zero_point)
        return scale, zero_point

    @staticmethod
    def backward(ctx, g_scale, g_zp):
        lower, upper, scale = ctx.saved_tensors
        neg = (lower < 0).to(g_scale.dtype)  # d min(lower, 0) / d lower
        live = (scale > 1e-6).to(g_scale.dtype)  # the 1e-6 floor kills the gradient
        if ctx.symmetric:
            a, b = -torch.clamp(lower, max=0), torch.clamp(upper, min=0)
            w = torch.where(a > b, torch.ones_like(a), torch.where(a == b, torch.full_like(a, 0.5), torch.zeros_like(a)))
            d = -w * 2.0 / ctx.qrange
        else:
            d = -torch.ones_like(lower) / ctx.qrange
        return g_scale * d * neg * live, None, None, None, None


@register_quantizer
class Quantizer(BaseQuantizer):
    TYPE = "PACT"

    def __init__(self, config):
        super(Quantizer, self).__init__(config)
        assert self.qdesc.target == QuantTarget.FEATURE, "PACT only support feature quantization"
        assert not self.qdesc.is_perchannel, "PACT no yet supports per-channel"
        self.init_alpha_value = config.QUANTIZER.PACT.ALPHA_VALUE

    def calc_qparams(self):
        if self.fake_fused:
            return self.scale, self.zero_point
        scale, zero_point = self.observer.calc_qparams()
        self.scale = self._broadcast_qparams(scale)
        self.zero_point = self._broadcast_qparams(zero_point)
        self.alpha = nn.Parameter(torch.Tensor([self.init_alpha_value]).to(self.device))
        return self.scale, self.zero_point

    def _qparams_preprocess(self, x):
        lower = -self.alpha if self.qdesc.qmin < 0 else torch.Tensor([0]).to(self.alpha.device)
        self.lower = lower
        # `lower` carries grad for symmetric ranges and the reference lets it reach alpha through
        # the scale; values come from the exact kernel (torch's GPU division is not correctly rounded)
        qmin, qmax = self.qdesc.qrange
        scale, zero_point = _QParamsFromBounds.apply(lower, self.alpha.detach(), qmin, qmax, self.is_symmetric)
        self.scale = self._broadcast_qparams(scale)
        self.zero_point = self._broadcast_qparams(zero_point)
        return self.scale, self.zero_point

    def _forward(self, x, scale, zero_point=None):
        x_clamp = torch.clamp(x, self.lower, self.alpha)  # differentiable w.r.t. alpha
        return STE.apply(x_clamp, scale, zero_point, self.qdesc, self.backend, self._out_dtype(x_clamp))
