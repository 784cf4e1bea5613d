"""Drop-in for the reference's pybind module `fake_quant`
(sparsebit/quantization/torch_extensions/export.cc:3-8, fake_quant_tensor.h:9-36):
the same four function names, argument order and return conventions, backed by
libsbq.so.  `sparsebit.quantization.quantizers.quant_tensor.fake_quant_kernel =
sparsebit_amd.fake_quant` is the whole native-boundary swap.

Differences, all supersets: fp16 / bf16 inputs are accepted (the reference
throws ValueTypeException on anything but fp32, common.cuh:45-49) and the launch
happens on the tensor's device (the reference ignores it -- no device guard).
The output is fp32 like the reference's.
"""
import torch

from . import lib as L
from . import ops


def _ch_geometry(data, ch_axis):
    if ch_axis < 0:
        ch_axis += data.dim()
    return ch_axis


def quant_pertensor_forward(data, scale, zero_point, qmin, qmax, rounding=0):
    L.require_device(data, scale, zero_point)
    if scale.numel() != 1:
        raise L.SbqError("per-tensor forward expects a single scale")
    return ops.fake_quant(data, scale, zero_point, qmin, qmax, 0, out_dtype=torch.float32, rounding=rounding)


def quant_perchannel_forward(data, scale, zero_point, qmin, qmax, ch_axis, rounding=0):
    L.require_device(data, scale, zero_point)
    ch_axis = _ch_geometry(data, ch_axis)
    if scale.numel() != data.shape[ch_axis]:
        raise L.SbqError("per-channel forward expects one scale per channel")
    if scale.numel() == 1:  # C == 1 degenerates to per tensor
        return quant_pertensor_forward(data, scale, zero_point, qmin, qmax, rounding)
    return ops.fake_quant(data, scale, zero_point, qmin, qmax, ch_axis, out_dtype=torch.float32, rounding=rounding)


def quant_pertensor_backward(data, scale, zero_point, grad_y, qmin, qmax, rounding=0):
    """-> [grad_x, grad_scale, grad_zero_point] shaped like the inputs (fake_quant_tensor.cu:147-149)."""
    gx, gs, gzp = ops.fake_quant_backward(data, grad_y, scale, zero_point, qmin, qmax, 0, True, True, rounding=rounding)
    return [gx, gs.reshape(scale.shape), gzp.reshape(zero_point.shape)]


def quant_perchannel_backward(data, scale, zero_point, grad_y, qmin, qmax, ch_axis, rounding=0):
    ch_axis = _ch_geometry(data, ch_axis)
    if scale.numel() == 1:
        return quant_pertensor_backward(data, scale, zero_point, grad_y, qmin, qmax, rounding)
    gx, gs, gzp = ops.fake_quant_backward(data, grad_y, scale, zero_point, qmin, qmax, ch_axis, True, True,
                                          rounding=rounding)
    return [gx, gs.reshape(scale.shape), gzp.reshape(zero_point.shape)]
