"""GPTQ 4- / 3- / 2-bit weight path (config 4): host-side mirror of
large_language_models/llama/quantization/utils/quant.py.

  quantize()                     quant.py:8-10
  Quantizer.find_params()        quant.py:43-132   (every branch: weights / activations, per channel /
                                                     per tensor, sym / asym, min-max / mse grid search)
  QuantLinear.pack / forward     quant.py:187-278
  Quant{4,3,2}Matmul             quant.py:281-403  -> sbq_vecquant{4,3,2}matmul

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
        self.maxq = torch.tensor(2 ** bit - 1)
        self.perchannel = perchannel
        self.sym = sym
        self.mse = mse
        self.norm = norm
        self.grid = grid
        self.maxshrink = maxshrink
        self.bit = bit

    @staticmethod
    def _geometry(shape, weight, groupsize):
        """Where a tensor's quantization channels live.  -> (channel axis, groups per channel, broadcast shape of the
        per-channel parameters).  Weights [out, in, ...]: axis 0, optionally `in` cut into groups of `groupsize`
        (parameters [out, groups, 1...]); activations: axis 1 of NCHW and of [tokens, features], the last axis of
        [batch, tokens, features] -- the layouts quant.py:55-68 / :111-132 spell out case by case."""
        rank = len(shape)
        if weight:
            groups = 1 if groupsize == -1 else shape[1] // groupsize
            lead = [shape[0], groups] if groups > 1 else [-1]
            return 0, groups, lead + [1] * (rank - 1)
        axis = 2 if rank == 3 else 1
        return axis, 1, [-1 if a == axis else 1 for a in range(rank)]

    def find_params(self, x, weight=False, groupsize=-1):
        """quant.py:43-132, every branch: weights [out, in] (optionally in groups) and activations of rank 4 / 3 / 2,
        per channel or per tensor, symmetric or not, min-max or the `mse` grid search.  The row statistics come from
        sbq_channel_stats, the grid search from sbq_gptq_mse_search; what is left are the reference's own handful of
        elementwise ops on [rows]-sized vectors."""
        self.maxq = self.maxq.to(x.device)
        if groupsize != -1:
            assert weight is True and x.shape[1] % groupsize == 0  # groupsize must be a divisor of infeatures
        axis, groups, param_shape = self._geometry(x.shape, weight, groupsize)
        n_channels = x.shape[axis]
        if not self.perchannel:
            rows = x.reshape(1, -1)  # one row: the whole tensor
        elif weight:
            rows = x.reshape(-1, groupsize) if groups > 1 else x.flatten(1)
        else:
            rows = x.movedim(axis, 0).flatten(1)
        rows = rows.contiguous()  # [rows, inner]: what the kernels read
        xmin, xmax, _ = ops.channel_stats(rows, 0, True)  # one read of x
        zero_t = torch.zeros_like(xmin)
        xmin = torch.minimum(xmin, zero_t)
        xmax = torch.maximum(xmax, zero_t)
        if self.sym:
            xmax = torch.maximum(torch.abs(xmin), xmax)
            xmin = torch.where(xmin < 0, -xmax, xmin)
        both0 = (xmin == 0) & (xmax == 0)
        xmin = torch.where(both0, torch.full_like(xmin, -1), xmin)
        xmax = torch.where(both0, torch.full_like(xmax, +1), xmax)
        scale = (xmax - xmin) / self.maxq
        zero = torch.full_like(scale, (self.maxq + 1) / 2) if self.sym else torch.round(-xmin / scale)
        if self.mse:
            scale, zero = scale.contiguous(), zero.contiguous()
            ops.gptq_mse_search(rows, xmin.contiguous(), xmax.contiguous(), int(self.maxq), self.sym, scale, zero,
                                self.norm, self.grid, int(self.maxshrink * self.grid))
        if not self.perchannel:  # the one pair serves every channel
            scale, zero = scale.repeat(n_channels), zero.repeat(n_channels)
        self.scale, self.zero = scale.reshape(param_shape), zero.reshape(param_shape)

    def quantize(self, x):
        if self.ready():
            return quantize(x, self.scale, self.zero, self.maxq)
        return x

    def enabled(self):
        return self.maxq > 0

    def ready(self):
        return torch.all(self.scale != 0)


MIN_GROUP = {4: 128, 3: 128, 2: 64}  # quant.py:153


def packed_rows(infeatures, bit):
    """rows of qweight: 3-bit levels come 32 to three words (quant.py:171-183)"""
    par = 3 if bit == 3 else 1
    return ceiling_div(infeatures * bit, 32 * par) * par


class QuantLinear(nn.Module):
    """2/3/4-bit packed linear layer; same buffers / state_dict layout as the reference's."""

    def __init__(self, infeatures, outfeatures, bit=4, groupsize=-1):
        super().__init__()
        assert bit in (2, 3, 4), "only support 2/3/4 bit now"
        if groupsize != -1:
            assert groupsize % MIN_GROUP[bit] == 0
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
        self.register_buffer("qweight", torch.zeros((packed_rows(infeatures, bit), outfeatures), dtype=torch.int))

    def pack(self, linear, scales, zeros):
        """quant.py:187-260: zeros' = zero*scale; intweight = round((w + zeros')/scale); then every
        output column becomes one little-endian bit stream over the rows of qweight, `bit` bits per
        input channel (8 nibbles or 16 crumbs per int32; for 3 bits, 32 levels per three int32 with
        the reference's two split levels being the ones that straddle a word).  Done with a handful
        of tensor ops on the weight's own device instead of the reference's numpy row loop."""
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
        bit = self.bit
        H = self.qweight.shape[0]
        level = intweight & (2 ** bit - 1)
        start = bit * torch.arange(self.infeatures, device=dev, dtype=torch.int64)  # stream position
        row, shift = start >> 5, start & 31
        lo = (level << shift.unsqueeze(1)) & 0xFFFFFFFF
        words = torch.zeros((H + 1, self.outfeatures), dtype=torch.int64, device=dev)
        words.index_add_(0, row, lo)  # disjoint bit fields: sum == or
        spill = shift + bit > 32
        if bool(spill.any()):
            hi = level[spill] >> (32 - shift[spill]).unsqueeze(1)
            words.index_add_(0, row[spill] + 1, hi)
        words = words[:H]
        words = torch.where(words >= 2 ** 31, words - 2 ** 32, words)  # two's complement int32
        self.qweight = words.to(torch.int32).contiguous()

    def forward(self, x):
        # fp32 inside like the reference (quant.py:262-278), result back in x.dtype
        fn = {2: Quant2Matmul, 3: Quant3Matmul, 4: Quant4Matmul}[self.bit]
        y = fn.apply(x.float(), self.qweight, self.scales.float(), self.zeros.float(), self.bias.float(), self.groupsize)
        return y.to(x.dtype)


def _quant_matmul(bits, input, qweight, scales, zeros, bias, groupsize):
    x_shape = list(input.shape)
    # y starts as the broadcast bias and is accumulated in place (quant.py:285-289)
    # .clone(): for batch 1 an expanded bias is already contiguous and would alias the buffer
    y = bias.to(input.dtype).expand(x_shape[:-1] + [bias.numel()]).clone(memory_format=torch.contiguous_format)
    ops.vecquantmatmul(bits, input.contiguous(), qweight, y, scales, zeros, 0 if groupsize == -1 else groupsize)
    return y


class Quant4Matmul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, qweight, scales, zeros, bias, groupsize=-1):
        return _quant_matmul(4, input, qweight, scales, zeros, bias, groupsize)

    @staticmethod
    def backward(ctx, grad):
        raise NotImplementedError("inference-only kernel (the reference's backward lives in alpaca-qlora)")


class Quant3Matmul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, qweight, scales, zeros, bias, groupsize=-1):
        return _quant_matmul(3, input, qweight, scales, zeros, bias, groupsize)

    @staticmethod
    def backward(ctx, grad):
        raise NotImplementedError("inference-only kernel")


class Quant2Matmul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, qweight, scales, zeros, bias, groupsize=-1):
        return _quant_matmul(2, input, qweight, scales, zeros, bias, groupsize)

    @staticmethod
    def backward(ctx, grad):
        raise NotImplementedError("inference-only kernel")


def quant_matmul_multi(input, layers):
    """[layer(input) for layer in layers] for up to 4 QuantLinear layers that read the SAME input (q / k / v of an
    attention block; gate + up of its MLP) with ONE mat-vec launch instead of one per layer (quant.py:262-278).
    Layers must share infeatures, bit width and group size; inference only, like Quant{4,3,2}Matmul."""
    first = layers[0]
    for ql in layers:
        if (ql.infeatures, ql.bit, ql.groupsize) != (first.infeatures, first.bit, first.groupsize):
            raise ValueError("quant_matmul_multi: layers must share infeatures, bit width and group size")
    x = input.reshape(-1, input.shape[-1]).float().contiguous()
    outs = []
    for ql in layers:
        out_shape = input.shape[:-1] + (ql.outfeatures,)
        outs.append(ql.bias.float().repeat(x.shape[0], 1).contiguous() if ql.bias is not None
                    else torch.zeros(x.shape[0], ql.outfeatures, dtype=torch.float32, device=x.device))
    gs = first.groupsize if first.groupsize != -1 else 0
    ops.vecquantmatmul_multi(first.bit, x, [ql.qweight for ql in layers], outs, [ql.scales for ql in layers],
                             [ql.zeros for ql in layers], gs)
    return [o.reshape(input.shape[:-1] + (ql.outfeatures,)).to(input.dtype) for o, ql in zip(outs, layers)]


class _KernelModule:
    """Stand-in for the reference's pybind module `cuda_kernel` (cuda/cuda_kernel.cpp:65-73): the six
    entry points with the reference's argument order, accumulating into `out` in place.  Assigning
    `utils.quant.cuda_kernel = sparsebit_amd.gptq.cuda_kernel` makes the reference's own
    Quant{2,3,4}Matmul run on these kernels."""

    @staticmethod
    def _run(bits, inp1, inp2, out, scales, zeros, group_size):
        if inp1.dim() < 2:
            raise RuntimeError("input1 must be with dimension > 2")
        ops.vecquantmatmul(bits, inp1.contiguous(), inp2, out, scales, zeros, group_size)

    def vecquant4matmul(self, inp1, inp2, out, scales, zeros):
        self._run(4, inp1, inp2, out, scales, zeros, 0)

    def vecgroupquant4matmul(self, inp1, inp2, out, scales, zeros, group_size):
        self._run(4, inp1, inp2, out, scales, zeros, group_size)

    def vecquant3matmul(self, inp1, inp2, out, scales, zeros):
        self._run(3, inp1, inp2, out, scales, zeros, 0)

    def vecgroupquant3matmul(self, inp1, inp2, out, scales, zeros, group_size):
        self._run(3, inp1, inp2, out, scales, zeros, group_size)

    def vecquant2matmul(self, inp1, inp2, out, scales, zeros):
        self._run(2, inp1, inp2, out, scales, zeros, 0)

    def vecgroupquant2matmul(self, inp1, inp2, out, scales, zeros, group_size):
        self._run(2, inp1, inp2, out, scales, zeros, group_size)


cuda_kernel = _KernelModule()
