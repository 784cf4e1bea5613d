"""QDQ-ONNX artifact without the `onnx` package (VERDICT r03 missing #4; quant_model.py:222-324).

CPU tests: the hand-written protobuf bytes (a) parse back with the module's own reader to the same q / scale /
zero_point / axis / bits, and (b) parse with GOOGLE's protobuf runtime against a descriptor of the ONNX messages built
here field by field from onnx.proto3's numbering -- an independent reader of the wire format."""
import os

import numpy as np
import pytest
import torch

from sparsebit_amd import onnx_qdq as X
from sparsebit_amd.export import QDQTensor, pack_int4


def _records():
    g = torch.Generator().manual_seed(0)
    q = torch.randint(-128, 128, (6, 5, 3, 3), generator=g).to(torch.int8)
    conv = QDQTensor(q, torch.rand(6, generator=g) + 0.01, torch.zeros(6, dtype=torch.int8), 0, 8, q.shape, True, False)
    q4 = torch.randint(0, 16, (4, 8), generator=g).to(torch.uint8)
    fc = QDQTensor(pack_int4(q4), torch.tensor(0.05), torch.tensor(7, dtype=torch.uint8), None, 4, q4.shape, False, True)
    qs = torch.randint(-8, 8, (3, 10), generator=g).to(torch.int8)
    lsq = QDQTensor(qs, torch.rand(3, generator=g) + 0.1, torch.tensor([-1, 0, 2], dtype=torch.int8), 0, 4, qs.shape, True, False)
    act = QDQTensor(None, torch.tensor(0.1), torch.tensor(-3, dtype=torch.int8), None, 8, (2, 3, 4, 4), True, False)
    act_pc = QDQTensor(None, torch.rand(3, generator=g), torch.tensor([1, 2, 3], dtype=torch.uint8), 1, 6, (2, 3, 4, 4), False, False)
    return {"conv.weight": conv, "fc.weight": fc, "lsq.weight": lsq}, {"x": act, "y": act_pc}, q4


def test_round_trip_through_own_reader(tmp_path):
    weights, acts, q4 = _records()
    path = os.path.join(str(tmp_path), "m.onnx")
    n = X.save_qdq_onnx(path, weights, acts)
    assert n == os.path.getsize(path) > 0
    back = X.load_qdq_onnx(path)
    assert back["opset"] == X.OPSET and sorted(back["weights"]) == sorted(weights) and sorted(back["activations"]) == sorted(acts)
    for name, rec in weights.items():
        got = back["weights"][name]
        assert torch.equal(got.q, rec.q) and torch.equal(got.scale, rec.scale.float()) and torch.equal(got.zero_point, rec.zero_point)
        assert got.axis == rec.axis and got.bits == rec.bits and got.shape == rec.shape
        assert got.signed == rec.signed and got.packed == rec.packed
    assert torch.equal(back["weights"]["fc.weight"].levels(), q4)
    for name, rec in acts.items():
        got = back["activations"][name]
        assert torch.equal(got["scale"], rec.scale.float()) and torch.equal(got["zero_point"], rec.zero_point)
        assert got["axis"] == rec.axis and got["bits"] == rec.bits and got["signed"] == rec.signed and got["shape"] == list(rec.shape)


def _onnx_descriptor_pool():
    """ModelProto and what it contains, from onnx.proto3's field numbers (the subset a QDQ file uses)"""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name="onnx_subset.proto", package="onnx", syntax="proto3")

    def msg(name, fields):
        m = fd.message_type.add(name=name)
        for fname, num, typ, rep, tname in fields:
            f = m.field.add(name=fname, number=num, type=typ, label=F.LABEL_REPEATED if rep else F.LABEL_OPTIONAL)
            if tname:
                f.type_name = ".onnx." + tname
        return m

    msg("TensorProto", [("dims", 1, F.TYPE_INT64, True, None), ("data_type", 2, F.TYPE_INT32, False, None),
                        ("name", 8, F.TYPE_STRING, False, None), ("raw_data", 9, F.TYPE_BYTES, False, None)])
    msg("AttributeProto", [("name", 1, F.TYPE_STRING, False, None), ("f", 2, F.TYPE_FLOAT, False, None),
                           ("i", 3, F.TYPE_INT64, False, None), ("type", 20, F.TYPE_INT32, False, None)])
    msg("NodeProto", [("input", 1, F.TYPE_STRING, True, None), ("output", 2, F.TYPE_STRING, True, None),
                      ("name", 3, F.TYPE_STRING, False, None), ("op_type", 4, F.TYPE_STRING, False, None),
                      ("attribute", 5, F.TYPE_MESSAGE, True, "AttributeProto")])
    msg("Dimension", [("dim_value", 1, F.TYPE_INT64, False, None)])
    msg("TensorShapeProto", [("dim", 1, F.TYPE_MESSAGE, True, "Dimension")])
    msg("TensorTypeProto", [("elem_type", 1, F.TYPE_INT32, False, None), ("shape", 2, F.TYPE_MESSAGE, False, "TensorShapeProto")])
    msg("TypeProto", [("tensor_type", 1, F.TYPE_MESSAGE, False, "TensorTypeProto")])
    msg("ValueInfoProto", [("name", 1, F.TYPE_STRING, False, None), ("type", 2, F.TYPE_MESSAGE, False, "TypeProto")])
    msg("GraphProto", [("node", 1, F.TYPE_MESSAGE, True, "NodeProto"), ("name", 2, F.TYPE_STRING, False, None),
                       ("initializer", 5, F.TYPE_MESSAGE, True, "TensorProto"), ("input", 11, F.TYPE_MESSAGE, True, "ValueInfoProto"),
                       ("output", 12, F.TYPE_MESSAGE, True, "ValueInfoProto")])
    msg("OperatorSetIdProto", [("domain", 1, F.TYPE_STRING, False, None), ("version", 2, F.TYPE_INT64, False, None)])
    msg("ModelProto", [("ir_version", 1, F.TYPE_INT64, False, None), ("producer_name", 2, F.TYPE_STRING, False, None),
                       ("graph", 7, F.TYPE_MESSAGE, False, "GraphProto"), ("opset_import", 8, F.TYPE_MESSAGE, True, "OperatorSetIdProto")])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("onnx.ModelProto"))


def test_bytes_parse_with_google_protobuf(tmp_path):
    pytest.importorskip("google.protobuf")
    weights, acts, _ = _records()
    path = os.path.join(str(tmp_path), "m.onnx")
    X.save_qdq_onnx(path, weights, acts)
    Model = _onnx_descriptor_pool()
    m = Model()
    with open(path, "rb") as f:
        m.ParseFromString(f.read())  # raises on malformed wire data
    assert m.ir_version == X.IR_VERSION and m.opset_import[0].version == X.OPSET and m.producer_name == "sparsebit_amd"
    g = m.graph
    ops = [n.op_type for n in g.node]
    assert ops.count("DequantizeLinear") == 5 and ops.count("QuantizeLinear") == 2
    inits = {t.name: t for t in g.initializer}
    t = inits["conv.weight_q"]
    assert list(t.dims) == [6, 5, 3, 3] and t.data_type == X.INT8
    assert np.array_equal(np.frombuffer(t.raw_data, np.int8).reshape(6, 5, 3, 3), weights["conv.weight"].q.numpy())
    assert np.array_equal(np.frombuffer(inits["conv.weight_scale"].raw_data, "<f4"), weights["conv.weight"].scale.numpy())
    assert inits["fc.weight_q"].data_type == X.UINT4 and list(inits["fc.weight_q"].dims) == [4, 8] and len(inits["fc.weight_q"].raw_data) == 16
    by_out = {n.output[0]: n for n in g.node}
    attrs = {a.name: a.i for a in by_out["conv.weight"].attribute}
    assert attrs == {"axis": 0, "bits": 8} and all(a.type == 2 for a in by_out["conv.weight"].attribute)
    assert {a.name: a.i for a in by_out["lsq.weight"].attribute} == {"axis": 0, "bits": 4}
    assert {a.name: a.i for a in by_out["fc.weight"].attribute} == {"bits": 4}  # per tensor: no axis
    assert {a.name: a.i for a in by_out["y_dq"].attribute} == {"axis": 1, "bits": 6}
    assert list(by_out["x_dq"].input) == ["x_q", "x_scale", "x_zero_point"] and list(by_out["x_q"].input)[0] == "x"
    assert [i.name for i in g.input] == ["x", "y"] and [d.dim_value for d in g.input[0].type.tensor_type.shape.dim] == [2, 3, 4, 4]
    assert inits["x_zero_point"].data_type == X.INT8 and np.frombuffer(inits["x_zero_point"].raw_data, np.int8)[0] == -3
