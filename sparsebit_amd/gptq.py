"""GPTQ 4-bit weight path (config 4): host-side mirror of
large_language_models/llama/quantization/utils/quant.py.

  quantize()                     quant.py:8-10
  Quantizer.find_params()        quant.py:43-132   (weight=True, perchannel, asymmetric,
                                                     mse=False -- what convert.py/test use)
  QuantLinear.pack / forward     quant.py:187-278
  Quant4Matmul                   quant.py:281-307  -> sbq_vecquant4matmul

find_params / quantize are a handful of tiny elementwise+rowwise torch ops on the
weight's own device (grouped min/max is served by sbq_channel_stats with
[out*groups, group] geometry); packing is an offline, one-time integer shuffle.
The per-token hot op is the mat-vec, which is the HIP kernel.
"""
import torch
import torch.nn as nn

from . import ops


def quantize(x, scale, zero, maxq):
    q = torch.clamp(torch.round(x / scale) + zero, 0, maxq)
    return scale * (q - zero)


def ceiling_div(x, y):
    return (x + y - 1) // y


class Quantizer(nn.Module):
    def __init__(self, shape=1):
        super(Quantizer, self).__init__()
        self.register_buffer("maxq", torch.tensor(0))
        self.register_buffer("scale", torch.zeros(shape))
        self.register_buffer("zero", torch.zeros(shape))

    def configure(self, bit, perchannel=False, sym=True, mse=False, norm=2.4, grid=100, maxshrink=0.8):
        if mse:
            raise NotImplementedError("GPTQ mse search is not on the MI355X hot path")
        self.maxq = torch.tensor(2 ** bit - 1)
        self.perchannel = perchannel
        self.sym = sym
        self.mse = mse
        self.bit = bit

    def find_params(self, x, weight=False, groupsize=-1):
        """Grouped asymmetric/symmetric min-max parameters of a [out, in] weight."""
        if not (weight and self.perchannel):
            raise NotImplementedError("only per-channel weight parameters are on the MI355X hot path")
        dev = x.device
        self.maxq = self.maxq.to(dev)
        shape = x.shape
        if groupsize != -1:
            assert x.shape[1] % groupsize == 0
            groups = x.shape[1] // groupsize
        else:
            groups = 1
        rows = x.reshape(-1, groupsize) if groups > 1 else x.flatten(1)
        xmin, xmax, _ = ops.channel_stats(rows.contiguous(), 0, True)  # [out*groups] each, one read
        zero_t = torch.zeros_like(xmin)
        xmin = torch.minimum(xmin, zero_t)
        xmax = torch.maximum(xmax, zero_t)
        if self.sym:
            xmax = torch.maximum(torch.abs(xmin), xmax)
            xmin = torch.where(xmin < 0, -xmax, xmin)
        both0 = (xmin == 0) & (xmax == 0)
        xmin = torch.where(both0, torch.full_like(xmin, -1), xmin)
        xmax = torch.where(both0, torch.full_like(xmax, +1), xmax)
        self.scale = (xmax - xmin) / self.maxq
        if self.sym:
            self.zero = torch.full_like(self.scale, (self.maxq + 1) / 2)
        else:
            self.zero = torch.round(-xmin / self.scale)
        if groups > 1:
            new_shape = [shape[0], groups] + [1] * (len(shape) - 1)
        else:
            new_shape = [-1] + [1] * (len(shape) - 1)
        self.scale = self.scale.reshape(new_shape)
        self.zero = self.zero.reshape(new_shape)

    def quantize(self, x):
        if self.ready():
            return quantize(x, self.scale, self.zero, self.maxq)
        return x

    def enabled(self):
        return self.maxq > 0

    def ready(self):
        return torch.all(self.scale != 0)


class QuantLinear(nn.Module):
    """4-bit packed linear layer; same buffers / state_dict layout as the reference's."""

    def __init__(self, infeatures, outfeatures, bit=4, groupsize=-1):
        super().__init__()
        if bit != 4:
            raise NotImplementedError("only the 4-bit mat-vec is on the MI355X hot path (2/3-bit: SURVEY.md 2)")
        if groupsize != -1:
            assert groupsize % 128 == 0
            assert infeatures % groupsize == 0
            groups = infeatures // groupsize
        else:
            groups = 1
        self.infeatures = infeatures
        self.outfeatures = outfeatures
        self.groups = groups
        self.groupsize = groupsize
        self.bit = bit
        shape = (outfeatures, groups, 1) if groups > 1 else (outfeatures, 1)
        self.register_buffer("zeros", torch.zeros(shape))
        self.register_buffer("scales", torch.zeros(shape))
        self.register_buffer("bias", torch.zeros(outfeatures))
        self.register_buffer("qweight", torch.zeros((ceiling_div(infeatures * 4, 32), outfeatures), dtype=torch.int))

    def pack(self, linear, scales, zeros):
        """quant.py:187-229 for bit == 4: zeros' = zero*scale; intweight = round((w + zeros')/scale);
        8 consecutive input channels per int32, low nibble first."""
        dev = linear.weight.device
        scales = scales.to(dev)
        zeros = zeros.to(dev)
        self.zeros = zeros * scales
        self.scales = scales.clone()
        self.bias = linear.bias.detach().clone() if linear.bias is not None else torch.zeros(self.outfeatures, device=dev)
        weight = linear.weight.data
        if self.groups > 1:
            weight = weight.view(self.outfeatures, self.groups, -1)
        intweight = torch.round((weight + self.zeros) / self.scales).to(torch.int64)
        intweight = intweight.reshape(self.outfeatures, self.infeatures).t().contiguous()  # [in, out]
        H = self.qweight.shape[0]
        pad = H * 8 - self.infeatures
        if pad:
            intweight = torch.cat([intweight, intweight.new_zeros(pad, self.outfeatures)], 0)
        nib = (intweight & 0xF).reshape(H, 8, self.outfeatures)
        shifts = (4 * torch.arange(8, device=dev, dtype=torch.int64)).reshape(1, 8, 1)
        words = (nib << shifts).sum(1)  # disjoint bit fields: sum == or
        words = torch.where(words >= 2 ** 31, words - 2 ** 32, words)  # two's complement int32
        self.qweight = words.to(torch.int32).contiguous()

    def forward(self, x):
        # fp32 inside like the reference (quant.py:262-278), result back in x.dtype
        y = Quant4Matmul.apply(x.float(), self.qweight, self.scales.float(), self.zeros.float(), self.bias.float(),
                               self.groupsize)
        return y.to(x.dtype)


class Quant4Matmul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, qweight, scales, zeros, bias, groupsize=-1):
        x_shape = list(input.shape)
        # y starts as the broadcast bias and is accumulated in place (quant.py:285-289)
        # .clone(): for batch 1 an expanded bias is already contiguous and would alias the buffer
        y = bias.to(input.dtype).expand(x_shape[:-1] + [bias.numel()]).clone(memory_format=torch.contiguous_format)
        ops.vecquant4matmul(input.contiguous(), qweight, y, scales, zeros, 0 if groupsize == -1 else groupsize)
        return y

    @staticmethod
    def backward(ctx, grad):
        raise NotImplementedError("inference-only kernel (the reference's backward lives in alpaca-qlora)")
