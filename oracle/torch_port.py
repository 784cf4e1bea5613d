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


def percentile_minmax_cpu(x, alpha):
    """percentile.py:16-46 per tensor (one row): two torch.kthvalue calls on the flattened data."""
    data = x.reshape(1, -1)
    neg_length = (data < 0).sum(-1)
    pos_length = (data >= 0).sum(-1)
    max_val = torch.zeros(1)
    min_val = torch.zeros(1)
    if pos_length[0] > 0:
        k = data.shape[1] - max(round(pos_length[0].item() * alpha), 0)
        max_val[0] = torch.kthvalue(data[0], k).values
    if neg_length[0] > 0:
        k = max(round(neg_length[0].item() * alpha), 1)
        min_val[0] = torch.kthvalue(data[0], k).values
    return min_val, max_val


def mse_qparams_cpu(x, qmin, qmax, symmetric=True):
    """mse.py:28-63 per tensor: 80 shrink candidates, each a fake-quant + squared error over the data."""
    x_f = x.reshape(-1)
    max_val, min_val = x_f.max(), x_f.min()
    best_scale = best_zp = None
    loss_min = 1e10
    for i in range(80):
        cur_min, cur_max = min_val * (1.0 - i * 0.01), max_val * (1.0 - i * 0.01)
        min_neg = torch.minimum(cur_min, torch.zeros(()))
        max_pos = torch.maximum(cur_max, torch.zeros(()))
        if symmetric:
            max_pos = torch.maximum(-min_neg, max_pos)
            scale = torch.maximum(max_pos * 2 / float(qmax - qmin), torch.tensor(1e-6))
            zp = torch.zeros(())
        else:
            scale = torch.maximum((max_pos - min_neg) / float(qmax - qmin), torch.tensor(1e-6))
            zp = torch.round(-min_neg / scale)
        x_dq = ort_fake_quant_cpu(x_f, scale, zp, qmin, qmax)
        loss = ((x_f - x_dq) ** 2).mean()
        if loss < loss_min:
            loss_min, best_scale, best_zp = loss, scale, zp
    return best_scale, best_zp


def mse_qparams_perchannel_cpu(x, qmin, qmax):
    """mse.py:28-63 per channel (symmetric): scale per row, loss = mean over the row (`mean(-1)`), strict < keeps the
    first best candidate per row.  Scales broadcast per ROW as the CUDA kernel indexes them
    (fake_quant_tensor.cu:181-186); the reference's CPU path mis-broadcasts a flat [C] scale."""
    max_val, min_val = x.max(1).values, x.min(1).values
    best_scale = torch.ones_like(max_val)
    loss_min = torch.full_like(max_val, 1e10)
    zero = torch.zeros_like(max_val)
    for i in range(80):
        cur_min, cur_max = min_val * (1.0 - i * 0.01), max_val * (1.0 - i * 0.01)
        min_neg = torch.minimum(cur_min, zero)
        max_pos = torch.maximum(torch.maximum(cur_max, zero), -min_neg)
        scale = torch.maximum(max_pos * 2 / float(qmax - qmin), torch.tensor(1e-6))
        x_dq = ort_fake_quant_cpu(x, scale.reshape(-1, 1), zero.reshape(-1, 1), qmin, qmax)
        loss = ((x - x_dq) ** 2).mean(-1)
        better = loss < loss_min
        loss_min = torch.where(better, loss, loss_min)
        best_scale = torch.where(better, scale, best_scale)
    return best_scale, zero


def l1_mask_cpu(w, ratio):
    """l1norm.py:18-26 unstructured: full sort of |w|, threshold at index n * ratio, strict >."""
    w_abs = w.abs()
    flat = w_abs.flatten().sort()[0]
    thresh = flat[min(int(flat.numel() * ratio), flat.numel() - 1)]
    return w_abs > thresh
