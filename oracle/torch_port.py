"""torch-CPU restatement of the reference's CPU fake-quant -- TEST INFRASTRUCTURE.

Used ONLY as bench.py's `cpu_baseline` (kind "port"): the reference's CPU Quantizer is
exactly these torch ops on host threads (sparsebit/quantization/quantizers/
quant_tensor.py:182-184, reached through Quantizer.forward -> uniform._forward ->
STE.apply -> ort_fake_quant), so timing them on the GPU box's host cores is timing the
reference's CPU path without needing /root/reference there.  Never imported by
sparsebit_amd/.
"""
import torch


def ort_fake_quant_cpu(x_f, scale, zero_point, qmin, qmax):
    zp = zero_point.round()
    x_q = torch.clamp((x_f / scale).round() + zp, qmin, qmax)
    x_dq = (x_q - zp) * scale
    return x_dq


def minmax_qparams_cpu(x, qmin, qmax):
    """minmax.py:14-25 + base.py:63-79, per-channel symmetric, on the host."""
    max_val = x.max(axis=1).values
    min_val = x.min(axis=1).values
    min_neg = torch.minimum(min_val, torch.zeros_like(min_val))
    max_pos = torch.maximum(max_val, torch.zeros_like(max_val))
    max_pos = torch.maximum(-min_neg, max_pos)
    scale = torch.maximum(max_pos * 2 / float(qmax - qmin), torch.tensor(1e-6))
    return scale, torch.zeros_like(scale)
