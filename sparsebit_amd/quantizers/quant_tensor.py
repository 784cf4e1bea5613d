"""Fake-quant dispatch + straight-through estimator on the HIP path.

Mirrors sparsebit/quantization/quantizers/quant_tensor.py: `fake_quant_kernel` is
the object with the reference pybind module's four functions (export.cc:3-8),
`ort_fake_quant` / `trt_fake_quant` / `fake_quant_factory` / `STE` /
`torch_fake_quant` keep their names, arguments and checks.  The GPU branch of the
reference is the only branch here: CPU tensors are rejected (no fallback).
"""
import numpy as np
import torch

from .. import fake_quant as fake_quant_kernel  # noqa: F401  (the name the reference's module exposes)
from .. import ops
from .. import plan
from ..common import Backend


def _same_device(x_f, scale, zero_point):
    assert (
        x_f.device == scale.device == zero_point.device
    ), "input, scale and zero_point of quantizer must be on same device!"


def ort_fake_quant(x_f, scale, zero_point, qdesc, out_dtype=None):
    """quant_tensor.py:159-185 (GPU branch).  bf16/fp16/fp32 in; fp32 out by default like
    the reference (it upcasts fp16 itself, :165-166)."""
    _same_device(x_f, scale, zero_point)
    qmin, qmax = qdesc.qrange
    return ops.fake_quant(x_f, scale, zero_point, qmin, qmax, qdesc.ch_axis, out_dtype=out_dtype or _default_out(x_f))


def _assert_symmetric(zero_point):
    """The reference's `assert abs(zero_point).sum() == 0` (quant_tensor.py:131-134) reads the device tensor on the
    host: a blocking D2H copy in front of a 3-11 us kernel, on every forward of every quantizer of a TensorRT-backend
    model (config 1).  The check is a property of the zero-point TENSOR, not of the call: it is made once per tensor
    object and in-place version (`_version` moves with every in-place write -- an optimizer step on a learnable
    zero point, BN fusion, load_state_dict's copy_) and the verdict rides on the tensor.  Same assertion, same
    message; 1000 forwards cost one host read (tests/test_gpu_r04.py::test_trt_forward_does_not_sync)."""
    ver = zero_point._version
    if getattr(zero_point, "_sbq_symmetric_checked", None) == ver:
        return
    assert abs(zero_point).sum() == 0, "tensorrt only support symmetric quant, but zp={}".format(zero_point)
    zero_point._sbq_symmetric_checked = ver


def trt_fake_quant(x_f, scale, zero_point, qdesc, out_dtype=None):
    """quant_tensor.py:128-156 (GPU branch): symmetric only."""
    _same_device(x_f, scale, zero_point)
    _assert_symmetric(zero_point)
    qmin, qmax = qdesc.qrange
    return ops.fake_quant(x_f, scale, zero_point, qmin, qmax, qdesc.ch_axis, out_dtype=out_dtype or _default_out(x_f))


def keep_input_dtype(flag=True):
    """Perf mode: return the dequantized tensor in the input dtype (bf16 in -> bf16 out,
    = RNE cast of the fp32 result, 4 B/element of HBM traffic instead of 6).  Default
    off: the reference always returns fp32.

    This is the PROCESS default; a quantizer's own `keep_input_dtype` attribute (None = follow the default, True /
    False = its own choice) overrides it, so two models in one process can differ."""
    plan.set_keep_default(flag)


def _default_out(x, keep=None):
    """output dtype of a quantizer call: `keep` is the quantizer's own setting (None: the process default)"""
    if keep is None:
        keep = plan.keep_default()
    return x.dtype if keep else torch.float32


fake_quant_factory = {
    Backend.VIRTUAL: ort_fake_quant,
    Backend.ONNXRUNTIME: ort_fake_quant,
    Backend.TENSORRT: trt_fake_quant,
}


class STE(torch.autograd.Function):
    """quant_tensor.py:74-125; backward through sbq_quant_*_backward.  (out_dtype: the calling quantizer's output
    dtype, None = the process default.)"""

    @staticmethod
    def forward(ctx, x, scale, zero_point, qdesc, backend, out_dtype=None):
        x_fq = fake_quant_factory[backend](x, scale, zero_point, qdesc, out_dtype)
        ctx.save_for_backward(x, scale, zero_point)
        ctx.qdesc = qdesc
        return x_fq

    @staticmethod
    def backward(ctx, gout):
        x, scale, zero_point = ctx.saved_tensors
        qdesc = ctx.qdesc
        qmin, qmax = qdesc.qrange
        need_gs = ctx.needs_input_grad[1]
        need_gzp = ctx.needs_input_grad[2]
        gx, gs, gzp = ops.fake_quant_backward(
            x, gout, scale, zero_point.float(), qmin, qmax, qdesc.ch_axis, need_gs, need_gzp, gx_dtype=x.dtype
        )
        if gs is not None:
            gs = gs.reshape(scale.shape)
        if gzp is not None:
            gzp = gzp.reshape(zero_point.shape)
        return gx, gs, gzp, None, None, None


def ste_fake_quant(x, scale, zero_point, qdesc, backend, out_dtype=None):
    """STE.apply, or -- when nothing asks for a gradient (PTQ calibration / evaluation, no_grad) -- the forward
    alone without an autograd node: the quantizer calls of an inference pass are host-bound."""
    if torch.is_grad_enabled() and (x.requires_grad or scale.requires_grad or zero_point.requires_grad):
        return STE.apply(x, scale, zero_point, qdesc, backend, out_dtype)
    return fake_quant_factory[backend](x, scale, zero_point, qdesc, out_dtype)


def trt_dqrange(scale, zero_point, qdesc):
    _assert_symmetric(zero_point)
    qmin, qmax = qdesc.qrange
    return (scale * qmin, scale * qmax)


def ort_dqrange(scale, zero_point, qdesc):
    qmin, qmax = qdesc.qrange
    return ((qmin - zero_point) * scale, (qmax - zero_point) * scale)


fake_qrange_factory = {
    Backend.VIRTUAL: ort_dqrange,
    Backend.ONNXRUNTIME: ort_dqrange,
    Backend.TENSORRT: trt_dqrange,
}


def torch_fake_quant(x_f, scale, zero_point, qdesc):
    """Export-only branch, kept on torch builtins so torch.onnx.export emits
    QuantizeLinear/DequantizeLinear exactly as with the reference (quant_tensor.py:220-249):
    the HIP kernel is never traced."""
    if qdesc._type.startswith("uint"):
        lower_bound, upper_bound = (0, 255)
    else:
        lower_bound, upper_bound = (-128, 127)
    if scale.numel() > 1:
        ch_axis = int(np.argmax(list(scale.shape)))
        scale = scale.reshape(-1).detach().to(x_f.device)
        zero_point = zero_point.reshape(-1).int().to(x_f.device)
        return torch.fake_quantize_per_channel_affine(x_f, scale, zero_point, ch_axis, lower_bound, upper_bound)
    if scale.numel() == 1:
        return torch.fake_quantize_per_tensor_affine(
            x_f, scale.item(), zero_point.int().item(), lower_bound, upper_bound
        )
    raise TypeError("scale / zeropoint is not allowed to be an empty tensor")
