"""Real quantized storage: what the fake-quant path turns into on disk.

The reference never materialises integer tensors itself; its formats are defined implicitly
by (a) the QDQ-ONNX graph `QuantModel.export_onnx` emits through `torch_fake_quant`
(sparsebit/quantization/quantizers/quant_tensor.py:220-249, quant_model.py:222-324:
QuantizeLinear / DequantizeLinear with int8 / uint8 tensors, per-channel `axis`, and a `bits`
attribute on both nodes for sub-8-bit) and (b) the GPTQ checkpoints (`QuantLinear.pack`,
large_language_models/llama/quantization/utils/quant.py:187-260; the int8 re-packing of
large_language_models/alpaca-qlora/convert_pack32topack8.py:4-34).

Here the integer tensor comes out of the same kernel launch as the fake-quant result
(`sbq_quant_*_forward` with SBQ_Q_I8 / SBQ_Q_I4), so `DequantizeLinear(q) == fake_quant(x)`
holds bit for bit by construction, and the GPTQ layouts convert with a couple of tensor ops.
"""
import torch

from . import ops


class QDQTensor:
    """One QuantizeLinear/DequantizeLinear pair's constants, ONNX convention:
    q = saturate(round(x / scale) + zero_point), dq = (q - zero_point) * scale.

      q           int8 (signed range) / uint8 levels, or packed nibbles when `packed`
      scale       fp32, [C] along `axis` (per channel) or 0-d
      zero_point  same integer type as q, same shape as scale
      axis        channel axis, None for per tensor
      bits        the `bits` attribute of quant_model.py:299-322 (8 for plain int8)
    """

    def __init__(self, q, scale, zero_point, axis, bits, shape, signed, packed):
        self.q, self.scale, self.zero_point = q, scale, zero_point
        self.axis, self.bits, self.shape, self.signed, self.packed = axis, bits, tuple(shape), signed, packed

    def levels(self):
        """integer levels in the tensor's shape (unpacks int4 storage)"""
        if not self.packed:
            return self.q
        return unpack_int4(self.q, self.signed).reshape(self.shape)

    def dequantize(self, out_dtype=None):
        """DequantizeLinear on the GPU (sbq_dequantize_linear): (q - zero_point) * scale, fp32 unless out_dtype"""
        return ops.dequantize_linear(self.q, self.scale, self.zero_point.to(torch.float32), shape=self.shape,
                                     ch_axis=0 if self.axis is None else self.axis, signed=self.signed,
                                     packed_int4=self.packed, out_dtype=out_dtype)

    def to(self, device):
        return QDQTensor(self.q.to(device), self.scale.to(device), self.zero_point.to(device), self.axis, self.bits,
                         self.shape, self.signed, self.packed)

    def state_dict(self, prefix=""):
        return {prefix + "q": self.q, prefix + "scale": self.scale, prefix + "zero_point": self.zero_point,
                prefix + "meta": torch.tensor([-1 if self.axis is None else self.axis, self.bits, int(self.signed),
                                               int(self.packed)] + list(self.shape), dtype=torch.int64)}

    @classmethod
    def from_state_dict(cls, sd, prefix=""):
        meta = [int(v) for v in sd[prefix + "meta"]]
        axis = None if meta[0] < 0 else meta[0]
        return cls(sd[prefix + "q"], sd[prefix + "scale"], sd[prefix + "zero_point"], axis, meta[1], meta[4:],
                   bool(meta[2]), bool(meta[3]))


@torch.no_grad()
def quantize_linear(quantizer, x, pack_int4=False, dequantized=True):
    """Materialise `x` with a calibrated quantizer -> (fake-quant result, QDQTensor); with
    dequantized=False the kernel writes the levels only and the first element is None.

    The integer container follows `torch_fake_quant` (quant_tensor.py:226-231): int8 for signed
    ranges, uint8 for unsigned, whatever the bit width; with pack_int4 a <= 4-bit tensor is stored
    two levels per byte straight from the kernel.
    """
    qdesc = quantizer.qdesc
    lo, hi = qdesc.qrange
    if hi - lo > 255:
        raise ValueError("QDQ-ONNX containers are 8 bits wide; %d-bit quantizers cannot be exported" % qdesc.bit)
    signed = lo < 0
    int_dtype = torch.int8 if signed else torch.uint8
    scale, zero_point = quantizer._qparams_preprocess(x)
    per_channel = scale.numel() > 1
    packed = bool(pack_int4) and hi - lo <= 15
    want = "int4" if packed else int_dtype
    if dequantized:
        dq, q = ops.fake_quant(x, scale, zero_point, lo, hi, qdesc.ch_axis, return_q=want)
    else:
        dq, q = None, ops.quantize_only(x, scale, zero_point, lo, hi, qdesc.ch_axis, return_q=want)
    rec = QDQTensor(q, scale.detach().reshape(-1).float() if per_channel else scale.detach().reshape(()).float(),
                    (zero_point.detach().round().reshape(-1) if per_channel else zero_point.detach().round().reshape(())).to(int_dtype),
                    qdesc.ch_axis if per_channel else None, qdesc.bit, x.shape, signed, packed)
    return dq, rec


def pack_int4(levels):
    """int8 / uint8 levels (already within 4 bits) -> uint8, element i in the low nibble of byte i//2"""
    flat = levels.reshape(-1).to(torch.uint8) & 0xF
    if flat.numel() % 2:
        raise ValueError("packed int4 needs an even number of elements")
    return (flat[0::2] | (flat[1::2] << 4)).contiguous()


def unpack_int4(packed, signed):
    """inverse of pack_int4 / of the kernel's SBQ_Q_I4 output -> int8 (signed) or uint8, flat"""
    b = packed.reshape(-1)
    out = torch.stack([b & 0xF, b >> 4], dim=1).reshape(-1)
    if signed:
        out = out.to(torch.int8)
        return torch.where(out > 7, out - 16, out)
    return out


# ---- GPTQ checkpoint layouts -------------------------------------------------------------------
def pack32_to_pack8(qweight):
    """convert_pack32topack8.py:13-29: every int32 row of qweight becomes four int8 rows, byte 0
    first -- [H, out] int32 -> [4H, out] int8 (for 4-bit weights: input channels 2r, 2r+1 in row r)."""
    if qweight.dtype != torch.int32:
        raise TypeError("weight in checkpoint must be int32!")
    shifts = torch.arange(4, device=qweight.device, dtype=torch.int32).reshape(1, 4, 1) * 8
    return ((qweight.unsqueeze(1) >> shifts) & 0xFF).reshape(-1, qweight.shape[1]).to(torch.int8)


def pack8_to_pack32(qweight8):
    """inverse of pack32_to_pack8"""
    if qweight8.dtype != torch.int8 or qweight8.shape[0] % 4:
        raise TypeError("pack8 weights are int8 with a multiple of 4 rows")
    b = (qweight8.to(torch.int64) & 0xFF).reshape(-1, 4, qweight8.shape[1])
    shifts = torch.arange(4, device=qweight8.device, dtype=torch.int64).reshape(1, 4, 1) * 8
    words = (b << shifts).sum(1)
    return torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32)


def convert_checkpoint_pack32_to_pack8(state_dict):
    """the loop of convert_pack32topack8.py:9-33 over a model state dict (every key containing
    'qweight'); returns a new dict, leaves the input untouched"""
    return {k: (pack32_to_pack8(v) if "qweight" in k else v) for k, v in state_dict.items()}


# ---- the QDQ-ONNX artifact ------------------------------------------------------------------------
@torch.no_grad()
def collect_qdq(model):
    """The constants of every enabled quantizer of a QuantOpr-style module tree (attributes input_quantizer /
    weight_quantizer / weight -- a reference QuantModel after plugin.install(), or any look-alike):
    -> (weights {"<module>.weight": QDQTensor with levels}, activations {"<module>.input": QDQTensor without q}).
    What quant_model.py:222-324 leaves in the exported graph, minus the float operators between the pairs."""
    weights, activations = {}, {}
    for name, m in model.named_modules():
        wq = getattr(m, "weight_quantizer", None)
        w = getattr(m, "weight", None)
        if wq is not None and getattr(wq, "is_enable", False) and isinstance(w, torch.Tensor):
            if wq.dims is None:
                wq.dims = w.dim()
            _, rec = quantize_linear(wq, w.detach(), dequantized=False)
            weights[name + ".weight"] = rec
        iq = getattr(m, "input_quantizer", None)
        if iq is not None and getattr(iq, "is_enable", False):
            qdesc = iq.qdesc
            lo, hi = qdesc.qrange
            signed = lo < 0
            scale, zero_point = iq._qparams_preprocess(None)
            per_channel = scale.numel() > 1
            int_dtype = torch.int8 if signed else torch.uint8
            activations[name + ".input"] = QDQTensor(
                None, scale.detach().reshape(-1).float() if per_channel else scale.detach().reshape(()).float(),
                (zero_point.detach().round().reshape(-1) if per_channel else zero_point.detach().round().reshape(())).to(int_dtype),
                qdesc.ch_axis if per_channel else None, qdesc.bit, (), signed, False)
    return weights, activations


def save_qdq_onnx(model_constants, path):
    """Write a QDQ-ONNX file: `model_constants` is a module tree (collect_qdq is run on it) or a
    (weights, activations) pair of {name: QDQTensor}.  Hand-written protobuf (sparsebit_amd/onnx_qdq.py): neither the
    `onnx` package nor a tracing exporter is involved.  -> bytes written."""
    from . import onnx_qdq

    if isinstance(model_constants, torch.nn.Module):
        model_constants = collect_qdq(model_constants)
    weights, activations = model_constants
    return onnx_qdq.save_qdq_onnx(path, weights, activations)


def load_qdq_onnx(path):
    from . import onnx_qdq

    return onnx_qdq.load_qdq_onnx(path)
