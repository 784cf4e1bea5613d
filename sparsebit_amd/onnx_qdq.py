"""QDQ-ONNX files without the `onnx` package.

The reference's export (sparsebit/quantization/quant_model.py:222-324) traces the model with
`torch.onnx.export` -- `torch_fake_quant` becomes QuantizeLinear / DequantizeLinear pairs -- and then, with
`extra_info=True`, re-opens the file with `onnx` and appends a `bits` attribute to both nodes of every pair
(:262-324) so that sub-8-bit quantizers survive in an 8-bit container.  Neither `onnx` nor a tracing exporter is
needed for the part of that artifact this path owns -- the CONSTANTS: integer levels, fp32 scales, integer zero
points, `axis`, `bits`.  An ONNX file is a protobuf message; the handful of message types involved
(ModelProto > GraphProto > NodeProto / TensorProto / AttributeProto / ValueInfoProto) are written and read back
here with ~100 lines of varint / length-delimited wire format.

    save_qdq_onnx(path, weights={name: QDQTensor}, activations={name: QDQTensor-like constants without q})
    load_qdq_onnx(path) -> {"weights": {name: QDQTensor}, "activations": {name: dict}, "opset": int}

Per weight:      initializers <name>_q / <name>_scale / <name>_zero_point and a DequantizeLinear node -> <name>
Per activation:  graph input <name>, QuantizeLinear + DequantizeLinear -> <name>_dq
`axis` is present on per-channel nodes only (ONNX's default 1 would be wrong for weights, axis 0); `bits` on every
node, as quant_model.py:299-322 writes it.  Packed int4 levels use ONNX's own INT4 / UINT4 tensor types (opset 21:
two elements per byte, element 2i in the low nibble -- the layout SBQ_Q_I4 writes).
"""
import struct

import numpy as np
import torch

from .export import QDQTensor

# TensorProto.DataType
FLOAT, UINT8, INT8, INT32, INT64, UINT4, INT4 = 1, 2, 3, 6, 7, 21, 22
_NP = {FLOAT: np.float32, UINT8: np.uint8, INT8: np.int8, INT32: np.int32, INT64: np.int64}
IR_VERSION, OPSET = 10, 21


# ---- protobuf wire format -----------------------------------------------------------------------
def _varint(v):
    v &= (1 << 64) - 1  # negative int64 -> two's complement, 10 bytes
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _key(field, wire):
    return _varint((field << 3) | wire)


def _int(field, v):
    return _key(field, 0) + _varint(int(v))


def _bytes(field, b):
    if isinstance(b, str):
        b = b.encode("utf-8")
    return _key(field, 2) + _varint(len(b)) + b


def _float(field, f):
    return _key(field, 5) + struct.pack("<f", f)


def _parse(buf):
    """one message -> [(field, wire, value)]; length-delimited values stay bytes"""
    out, i, n = [], 0, len(buf)
    while i < n:
        k, i = _read_varint(buf, i)
        field, wire = k >> 3, k & 7
        if wire == 0:
            v, i = _read_varint(buf, i)
        elif wire == 2:
            ln, i = _read_varint(buf, i)
            v = bytes(buf[i:i + ln])
            i += ln
        elif wire == 5:
            v = struct.unpack("<f", buf[i:i + 4])[0]
            i += 4
        elif wire == 1:
            v = struct.unpack("<d", buf[i:i + 8])[0]
            i += 8
        else:
            raise ValueError("unsupported wire type %d" % wire)
        out.append((field, wire, v))
    return out


def _read_varint(buf, i):
    v = s = 0
    while True:
        b = buf[i]
        i += 1
        v |= (b & 0x7F) << s
        s += 7
        if not b & 0x80:
            return v, i


def _signed(v):
    return v - (1 << 64) if v >= 1 << 63 else v


# ---- ONNX messages ----------------------------------------------------------------------------------
def _tensor(name, data_type, dims, raw):
    msg = b"".join(_int(1, d) for d in dims) + _int(2, data_type) + _bytes(8, name) + _bytes(9, raw)
    return msg


def _attr_int(name, v):
    return _bytes(1, name) + _int(3, v) + _int(20, 2)  # AttributeProto.INT


def _node(op_type, inputs, outputs, name, attrs):
    msg = b"".join(_bytes(1, s) for s in inputs) + b"".join(_bytes(2, s) for s in outputs) + _bytes(3, name) + _bytes(4, op_type)
    return msg + b"".join(_bytes(5, a) for a in attrs)


def _value_info(name, elem_type, shape):
    dims = b"".join(_bytes(1, _int(1, d)) for d in shape)
    tensor_type = _int(1, elem_type) + _bytes(2, dims)
    return _bytes(1, name) + _bytes(2, _bytes(1, tensor_type))


def _np(t):
    return t.detach().cpu().contiguous().numpy()


def _constants(name, rec):
    """initializers of one record's scale / zero point (and levels) -> (list of TensorProto bytes, q elem type)"""
    signed = bool(rec.signed)
    q_type = (INT4 if signed else UINT4) if rec.packed else (INT8 if signed else UINT8)
    # DequantizeLinear requires x and x_zero_point to share a type: 4-bit levels take a 4-bit zero point (two per byte,
    # low nibble first -- ONNX's packing of INT4 / UINT4 raw_data -- the last byte padded with 0)
    z_type = q_type
    per_channel = rec.axis is not None
    scale = _np(rec.scale.float()).reshape(-1)
    zp = _np(rec.zero_point).reshape(-1).astype(np.int8 if signed else np.uint8)
    sdims = [scale.size] if per_channel else []
    if rec.packed:
        nib = zp.astype(np.uint8) & 0xF
        if nib.size % 2:
            nib = np.concatenate([nib, np.zeros(1, np.uint8)])
        zraw = (nib[0::2] | (nib[1::2] << 4)).astype(np.uint8).tobytes()
    else:
        zraw = zp.tobytes()
    ts = [_tensor(name + "_scale", FLOAT, sdims, scale.astype("<f4").tobytes()),
          _tensor(name + "_zero_point", z_type, sdims, zraw)]
    return ts, q_type


def save_qdq_onnx(path, weights=None, activations=None, producer="sparsebit_amd"):
    """weights: {name: QDQTensor with levels}; activations: {name: QDQTensor whose q is None (constants only; `shape`
    is the activation's shape)}.  Returns the number of bytes written."""
    nodes, inits, inputs, outputs = [], [], [], []
    for name, rec in (weights or {}).items():
        ts, q_type = _constants(name, rec)
        q = _np(rec.q)
        inits.append(_tensor(name + "_q", q_type, list(rec.shape), q.tobytes()))
        inits += ts
        attrs = ([_attr_int("axis", rec.axis)] if rec.axis is not None else []) + [_attr_int("bits", rec.bits)]
        nodes.append(_node("DequantizeLinear", [name + "_q", name + "_scale", name + "_zero_point"], [name],
                           name + "/DequantizeLinear", attrs))
        outputs.append(_value_info(name, FLOAT, rec.shape))
    for name, rec in (activations or {}).items():
        ts, _ = _constants(name, rec)
        inits += ts
        attrs = ([_attr_int("axis", rec.axis)] if rec.axis is not None else []) + [_attr_int("bits", rec.bits)]
        nodes.append(_node("QuantizeLinear", [name, name + "_scale", name + "_zero_point"], [name + "_q"],
                           name + "/QuantizeLinear", attrs))
        nodes.append(_node("DequantizeLinear", [name + "_q", name + "_scale", name + "_zero_point"], [name + "_dq"],
                           name + "/DequantizeLinear", attrs))
        inputs.append(_value_info(name, FLOAT, rec.shape))
        outputs.append(_value_info(name + "_dq", FLOAT, rec.shape))
    graph = b"".join(_bytes(1, n) for n in nodes) + _bytes(2, "sparsebit_amd_qdq") + b"".join(_bytes(5, t) for t in inits)
    graph += b"".join(_bytes(11, v) for v in inputs) + b"".join(_bytes(12, v) for v in outputs)
    model = _int(1, IR_VERSION) + _bytes(2, producer) + _bytes(7, graph) + _bytes(8, _bytes(1, "") + _int(2, OPSET))
    with open(path, "wb") as f:
        f.write(model)
    return len(model)


def load_qdq_onnx(path):
    """Parse a file written by save_qdq_onnx (or any ONNX file whose QuantizeLinear / DequantizeLinear constants are
    initializers with raw_data) back into records.  No `onnx` import."""
    with open(path, "rb") as f:
        model = _parse(f.read())
    graph = next(v for f_, w, v in model if f_ == 7)
    opset = None
    for f_, w, v in model:
        if f_ == 8:
            for g_, _, x in _parse(v):
                if g_ == 2:
                    opset = x
    tensors, nodes, vinfo = {}, [], {}
    for f_, w, v in _parse(graph):
        if f_ == 5:
            dims, dt, name, raw = [], None, None, b""
            for g_, _, x in _parse(v):
                if g_ == 1:
                    dims.append(_signed(x))
                elif g_ == 2:
                    dt = x
                elif g_ == 8:
                    name = x.decode()
                elif g_ == 9:
                    raw = x
            tensors[name] = (dt, dims, raw)
        elif f_ == 1:
            n = {"input": [], "output": [], "attrs": {}}
            for g_, _, x in _parse(v):
                if g_ == 1:
                    n["input"].append(x.decode())
                elif g_ == 2:
                    n["output"].append(x.decode())
                elif g_ == 4:
                    n["op_type"] = x.decode()
                elif g_ == 5:
                    a = {g2: x2 for g2, _, x2 in _parse(x)}
                    n["attrs"][a[1].decode()] = _signed(a[3])
            nodes.append(n)
        elif f_ in (11, 12):
            name, shape = None, []
            for g_, _, x in _parse(v):
                if g_ == 1:
                    name = x.decode()
                elif g_ == 2:
                    tt = dict((a, c) for a, _, c in _parse(x))[1]
                    for a, _, c in _parse(tt):
                        if a == 2:
                            shape = [_signed(dict((p, q) for p, _, q in _parse(d)).get(1, 0)) for _, _, d in _parse(c)]
            vinfo[name] = shape

    def array(name):
        dt, dims, raw = tensors[name]
        if dt in (INT4, UINT4):
            return torch.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()), dims, dt
        a = np.frombuffer(raw, dtype=np.dtype(_NP[dt]).newbyteorder("<")).copy()
        return torch.from_numpy(a.reshape(dims) if dims else a.reshape(())), dims, dt

    out = {"weights": {}, "activations": {}, "opset": opset}
    for n in nodes:
        if n["op_type"] != "DequantizeLinear":
            continue
        qn, sn, zn = n["input"]
        scale, sdims, _ = array(sn)
        zp, zdims, zt = array(zn)
        if zt in (INT4, UINT4):  # two zero points per byte, low nibble first -> one int8 / uint8 value each
            count = 1
            for d in zdims:
                count *= d
            b = zp.numpy()
            nib = np.stack([b & 0xF, b >> 4], axis=1).reshape(-1)[:count].astype(np.int16)
            if zt == INT4:
                nib = np.where(nib >= 8, nib - 16, nib)
            zp = torch.from_numpy(nib.astype(np.int8 if zt == INT4 else np.uint8).reshape(zdims if zdims else ()))
        axis = n["attrs"].get("axis") if sdims else None
        bits = n["attrs"].get("bits", 8)
        signed = zt in (INT8, INT4)
        if qn in tensors:  # stored levels: a weight
            q, qdims, qt = array(qn)
            signed = qt in (INT8, INT4)  # the levels' own type decides (the zero point shares it in a valid file)
            out["weights"][n["output"][0]] = QDQTensor(q, scale, zp, axis, bits, qdims, signed, qt in (INT4, UINT4))
        else:  # produced by the QuantizeLinear in front of it: an activation's constants
            name = qn[:-2] if qn.endswith("_q") else qn
            out["activations"][name] = {"scale": scale, "zero_point": zp, "axis": axis, "bits": bits, "signed": signed,
                                        "shape": vinfo.get(name, [])}
    return out
