"""LSQ quantizer (mirrors sparsebit/quantization/quantizers/lsq.py:13-76).

Init: scale = 2 * mean|x| / sqrt(qmax) from ONE fused min/max/sum|x| reduction of the
cached data (the reference materialises a channel-first copy and makes three torch
passes); forward: the same STE kernel with the LSQ gradient scaling.  A fused
mask + LSQ forward for sparse QAT is `forward_masked`.
"""
import math
import warnings

import torch
import torch.nn as nn

from . import Quantizer as BaseQuantizer
from . import register_quantizer
from .. import dist as sbq_dist
from .. import ops
from ..common import Backend
from ..registry import impl_type
from .quant_tensor import STE, _default_out


class gs_scaling(torch.autograd.Function):
    """identity forward, grad * ratio backward (lsq.py:13-21)"""

    @staticmethod
    def forward(ctx, x, ratio):
        ctx.ratio = ratio
        return x

    @staticmethod
    def backward(ctx, grad):
        return grad * ctx.ratio, None


class LsqSTE(torch.autograd.Function):
    """The whole LSQ quantizer step as ONE autograd node on the raw parameters: |scale|, clamp(zero_point),
    the gradient scaling and sign(scale) happen inside the kernels (sbq_quant_lsq_forward / _backward) instead
    of as abs / clamp / gs_scaling tensor ops and autograd nodes around the STE (lsq.py:13-21,61-76)."""

    @staticmethod
    def forward(ctx, x, scale, zero_point, qdesc, ratio, out_dtype=None):
        ctx.save_for_backward(x, scale, zero_point)
        ctx.qdesc, ctx.ratio = qdesc, ratio
        qmin, qmax = qdesc.qrange
        return ops.lsq_fake_quant(x, scale.detach(), zero_point.detach(), qmin, qmax, qdesc.ch_axis,
                                  out_dtype=out_dtype or _default_out(x))

    @staticmethod
    def backward(ctx, gout):
        x, scale, zero_point = ctx.saved_tensors
        qmin, qmax = ctx.qdesc.qrange
        gx, gs = ops.lsq_fake_quant_backward(x, gout, scale, zero_point, qmin, qmax, ctx.qdesc.ch_axis,
                                             ctx.needs_input_grad[1], ctx.ratio, gx_dtype=x.dtype)
        if gs is not None:
            gs = gs.reshape(scale.shape)
        return gx if ctx.needs_input_grad[0] else None, gs, None, None, None, None


@register_quantizer
class Quantizer(BaseQuantizer):
    TYPE = "LSQ"

    def __init__(self, config):
        super(Quantizer, self).__init__(config)
        self.init_params = False  # LSQ initialises from calibration data

    def calc_qparams(self):
        if self.fake_fused:
            return self.scale, self.zero_point
        if not self.init_params:
            shards = self.observer._shards()
            ch_axis, perch = self.qdesc.ch_axis, self.is_perchannel
            # lsq.py:39-43 needs the global minimum; lsq.py:44-47 the per-channel mean|x|:
            # one pass gives both (per-channel stats also yield the global min)
            mn = ab = None
            n_local = 0
            C = shards[0].shape[ch_axis] if perch else 1
            for x in shards:
                a, _, s = ops.channel_stats(x, ch_axis, perch, want_min=True, want_max=False, want_abssum=True)
                mn = a if mn is None else torch.minimum(mn, a)
                ab = s if ab is None else ab + s
                n_local += x.numel() // C
            if sbq_dist.active():
                mn, _ = sbq_dist.allreduce_minmax(mn, mn)
                sbq_dist.allreduce_sum_(ab)
            n = sbq_dist.allreduce_count(n_local)
            if bool(mn.min() < 0) and not self.qdesc.is_symmetric:
                warnings.warn("Found data less than 0, reset quantizer scheme as symmetric")
                self.qdesc.set_symmetric(True)
            scale = ops.lsq_init_scale(ab, n, self.qdesc.qmax)
            if not perch:
                scale = scale.reshape(())
            self.observer.data_cache.reset()
            self.scale = nn.Parameter(self._broadcast_qparams(scale.to(self.device)))
            self.zero_point = self._broadcast_qparams(torch.zeros_like(self.scale))
            self.init_params = True
        return self.scale, self.zero_point

    def _qparams_preprocess(self, x):
        if self.export_onnx:
            # the reference rebuilds both tensors through numpy (lsq.py:53-63) so that the ONNX tracer sees plain
            # constants instead of the Parameter's abs / clamp graph; a detached clone on the device is the same
            # values with no graph and no host round trip
            with torch.no_grad():
                scale = self.scale.detach().abs().clone().to(self.device)
                zero_point = torch.clamp(self.zero_point.detach(), self.qdesc.qmin, self.qdesc.qmax).clone().to(self.device)
            return scale, zero_point
        scale = self.scale.abs()
        zero_point = torch.clamp(self.zero_point, self.qdesc.qmin, self.qdesc.qmax)
        return scale, zero_point

    def forward(self, x):
        # plain LSQ on the ORT-style backends: one fused node (zero_point is a buffer here; LSQ+ learns it and
        # keeps the generic route, as does the TensorRT backend with its zero-point assertion)
        if (self.is_enable and not self.export_onnx and impl_type(self) is Quantizer and self.backend != Backend.TENSORRT
                and not self.zero_point.requires_grad and x.is_cuda):
            pre = self._pregrouped
            if pre is not None and x is pre[0]:
                return pre[1]
            if not (torch.is_grad_enabled() and (x.requires_grad or self.scale.requires_grad)):
                # nothing asks for a gradient: the forward alone, no autograd node -- through the launch plan
                # (sparsebit_amd.plan) when there is one for this input
                p = self._plans.lookup(self, x, lsq=True)
                if p is not None:
                    return p(x)
                qmin, qmax = self.qdesc.qrange
                return ops.lsq_fake_quant(x, self.scale.detach(), self.zero_point, qmin, qmax, self.qdesc.ch_axis,
                                          out_dtype=self._out_dtype(x))
            return LsqSTE.apply(x, self.scale, self.zero_point, self.qdesc, self._gs_ratio(x), self._out_dtype(x))
        return super().forward(x)

    def _gs_ratio(self, x):
        if self.is_perchannel:
            num_perchannel = x.numel() / x.shape[self.qdesc.ch_axis]
            return 1.0 / math.sqrt(num_perchannel * self.qdesc.qmax)
        return 1.0 / math.sqrt(x.numel() * self.qdesc.qmax)

    def _forward(self, x, scale, zero_point):
        ratio = self._gs_ratio(x)
        scale = gs_scaling.apply(scale, ratio)
        if zero_point.requires_grad:  # only LSQ+ learns its zero point (lsq_plus.py:90-92)
            zero_point = gs_scaling.apply(zero_point, ratio)
        return STE.apply(x, scale, zero_point, self.qdesc, self.backend, self._out_dtype(x))

    def forward_masked(self, x, mask=None, thresh=None, out_dtype=None):
        """Inference-side fused `quantizer(x * mask)`: one kernel, one read of x
        (sparse/modules/conv.py:40 + this quantizer).  mask: torch.bool like x, or
        thresh: 0-d tensor with keep = |x| > thresh."""
        if not self.is_enable:
            return x * mask if mask is not None else torch.where(x.abs() > thresh, x, torch.zeros_like(x))
        scale, zero_point = self._qparams_preprocess(x)
        qmin, qmax = self.qdesc.qrange
        return ops.fake_quant(x, scale.detach(), zero_point, qmin, qmax, self.qdesc.ch_axis,
                              out_dtype=out_dtype or torch.float32, mask=mask, thresh=thresh)
