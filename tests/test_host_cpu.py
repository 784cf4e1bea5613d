"""CPU tests: host logic, C-ABI surface, registries, no-fallback guarantee, build."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    from sparsebit_amd import build

    return build.build(verbose=False)


def test_library_exports_every_symbol_of_the_header(built):
    from sparsebit_amd import lib

    names = lib.header_symbols()
    assert len(names) >= 25 and "sbq_quant_perchannel_forward" in names
    assert set(names) == set(lib._SIGNATURES), set(names) ^ set(lib._SIGNATURES)
    raw = ctypes.CDLL(built)
    for n in names:
        getattr(raw, n)  # AttributeError if not exported
    out = subprocess.run(["nm", "-D", "--defined-only", built], stdout=subprocess.PIPE, text=True).stdout
    exported = set(re.findall(r" T (sbq_[a-z0-9_]+)", out))
    assert exported == set(names), "exported C symbols and include/sbq.h differ: %s" % (exported ^ set(names))
    assert lib.load().sbq_version() == 100


def test_status_strings_and_argument_validation_without_gpu(built):
    """Validation happens before any launch, so it is testable on a GPU-less host."""
    from sparsebit_amd import lib

    l = lib.load()
    assert l.sbq_strerror(0) == b"ok"
    assert b"Invalid dtype" in l.sbq_strerror(1) and b"Tensor is empty" in l.sbq_strerror(2)
    buf = (ctypes.c_float * 16)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    args = dict(x=p, y=p, s=p, z=p)
    f = l.sbq_quant_perchannel_forward
    assert f(p, 9, p, 0, None, 0, p, p, 1, 2, 8, -128, 127, 0, None) == 1  # dtype
    assert f(p, 0, p, 2, None, 0, p, p, 1, 2, 8, -128, 127, 0, None) == 1  # y dtype must be f32 or x's
    assert f(p, 0, p, 0, None, 0, p, p, 1, 0, 8, -128, 127, 0, None) == 2  # empty
    assert f(None, 0, p, 0, None, 0, p, p, 1, 2, 8, -128, 127, 0, None) == 3  # null
    assert f(p, 0, p, 0, None, 1, p, p, 1, 2, 8, -128, 127, 0, None) == 3  # q requested, NULL q
    assert f(p, 0, p, 0, None, 0, p, p, 1, 2, 8, 127, -128, 0, None) == 4  # qmin > qmax
    assert f(p, 0, p, 0, p, 1, p, p, 1, 2, 8, -32768, 32767, 0, None) == 4  # int8 storage too narrow
    assert f(p, 0, p, 0, None, 0, p, p, 1, 2, 8, -128, 127, 7, None) == 4  # rounding mode
    assert l.sbq_mask_quant_forward(p, 0, p, 0, None, 0, None, None, p, p, 1, 2, 8, -128, 127, 0, None) == 4
    assert l.sbq_vecquant4matmul(p, p, p, p, p, 1, 128, 4, 64, p, 1 << 20, None) == 4  # group % 128
    assert l.sbq_dequantize_linear(p, 7, 1, p, 0, p, p, 1, 4, 8, None) == 1  # unknown level type
    assert l.sbq_dequantize_linear(p, 1, 1, None, 0, p, p, 1, 4, 8, None) == 3
    assert l.sbq_quant_perchannel_forward(p, 2, None, 2, None, 0, p, p, 1, 4, 8, -8, 7, 0, None) == 3  # y NULL needs q
    assert l.sbq_quant_lsq_forward(p, 9, p, 0, None, p, p, 1, 4, 8, -8, 7, None) == 1  # dtype
    assert l.sbq_quant_lsq_forward(p, 0, p, 0, None, None, p, 1, 4, 8, -8, 7, None) == 3  # scale NULL
    assert l.sbq_quant_lsq_backward(p, p, 0, p, 0, p, p, p, 1, 4, 8, -8, 7, 0.5, None, 0, None) == 3  # gs wants a workspace
    assert l.sbq_vecquant3matmul(p, p, p, p, p, 1, 128, 4, 64, p, 1 << 20, None) == 4
    assert l.sbq_vecquant2matmul(p, p, p, p, p, 1, 128, 4, 32, p, 1 << 20, None) == 4  # group % 64
    assert l.sbq_percentile_rows(p, 0, 4, 20000, 0.001, p, p, None) == 4  # row too long for the LDS path
    assert l.sbq_channel_stats(p, 0, 1, 1, 16, p, p, None, p, 0, None) == 5  # workspace too small
    assert l.sbq_stats_workspace_bytes(1, 4096, 4096) == 4096 * 16
    assert l.sbq_mse_workspace_bytes(1, 4096, 4096) == 4096 * 80 * 8  # one chunk per channel: no fold levels
    assert l.sbq_gptq_workspace_bytes(1, 4096, 4096) > 0
    # selection workspace = [whole-tensor engine | fixed-digit passes]: the engine's share is the same for every C and
    # the fixed-digit histograms (int64 [C][n_sel][2048] + state + counts) come behind it, never on top of it
    w1, w64 = l.sbq_radix_select_workspace_bytes(1, 2), l.sbq_radix_select_workspace_bytes(64, 2)
    fixed = lambda C: C * 2 * 2048 * 8 + C * 2 * 16 + C * 16 + 64
    assert w1 - fixed(1) == w64 - fixed(64) > 128 * 1024
    assert l.sbq_radix_select_workspace_bytes(0, 2) == 0 and l.sbq_radix_select_workspace_bytes(1, 3) == 0
    assert l.sbq_group_kth_workspace_bytes(3) == 3 * l.sbq_group_kth_workspace_bytes(1) > 0


def test_no_cpu_fallback_in_product_path():
    """CPU tensors must be refused loudly, and nothing under sparsebit_amd/ may import oracle/."""
    from sparsebit_amd import lib, ops
    from sparsebit_amd.config import quantizer_config, sparser_config
    from sparsebit_amd.quantizers import build_quantizer
    from sparsebit_amd.sparsers import build_sparser

    with pytest.raises(lib.SbqError, match="no CPU fallback"):
        ops.fake_quant(torch.randn(4, 8), torch.ones(4), torch.zeros(4), -128, 127)
    with pytest.raises(lib.SbqError):
        ops.channel_stats(torch.randn(4, 8))
    q = build_quantizer(quantizer_config("per-tensor-symmetric", 8))
    q.enable_quant()
    q.backend = __import__("sparsebit_amd.common", fromlist=["Backend"]).Backend.VIRTUAL
    with pytest.raises(lib.SbqError):
        q(torch.randn(8))
    if not torch.cuda.is_available():
        q.update_observer(torch.randn(4, 8))
        with pytest.raises((lib.SbqError, AssertionError, RuntimeError)):
            q.calc_qparams()
        with pytest.raises(lib.SbqError):
            build_sparser(sparser_config(0.5)).calc_mask(torch.randn(16))
    for dirpath, _, files in os.walk(os.path.join(ROOT, "sparsebit_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f
                assert "libsbq_oracle" not in src, f


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from sparsebit_amd import lib

    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="libsbq.so not found"):
        lib.load()


def test_registries_and_descriptor_match_reference_contract():
    from sparsebit_amd import observers, quantizers, sparsers
    from sparsebit_amd.config import quantizer_config
    from sparsebit_amd.quantizers.quant_descriptor import QuantDescriptor

    assert {"uniform", "lsq"} <= set(quantizers.QUANTIZERS_MAP)
    assert {"minmax", "mse", "percentile"} <= set(observers.OBSERVERS_MAP)
    assert "l1norm" in sparsers.SPARSERS_MAP
    # quant_descriptor.py:28-34: symmetric [-2^(b-1), 2^(b-1)-1], affine [0, 2^b-1]
    for scheme, bit, lo, hi in (("per-channel-symmetric", 8, -128, 127), ("per-tensor-affine", 8, 0, 255),
                                ("per-channel-symmetric", 4, -8, 7), ("per-tensor-affine", 4, 0, 15)):
        d = QuantDescriptor(quantizer_config(scheme, bit))
        assert d.qrange == (lo, hi) and d.ch_axis == 0 and d.bs_axis is None
    assert QuantDescriptor(quantizer_config("per-tensor-affine", 8, target="feature", layout="NCHW")).ch_axis == 1
    assert QuantDescriptor(quantizer_config("per-tensor-affine", 8, target="feature", layout="NLC")).ch_axis == 2
    d = QuantDescriptor(quantizer_config("per-channel-affine", 4))
    d.set_symmetric(True)
    assert d.qrange == (-8, 7) and d.scheme == torch.per_channel_symmetric
    d.set_bit(8)
    assert d.qrange == (-128, 127)

    # a later registration overwrites, like the reference's maps
    @quantizers.register_quantizer
    class Mine(quantizers.Quantizer):
        TYPE = "MyQ"

    assert quantizers.QUANTIZERS_MAP["myq"] is Mine
    del quantizers.QUANTIZERS_MAP["myq"]
    q = quantizers.build_quantizer(quantizer_config("per-channel-symmetric", 8, quantizer="LSQ"))
    assert q.TYPE == "LSQ" and sorted(q.state_dict()) == ["observer.max_val", "observer.min_val", "scale", "zero_point"]
    assert not q.is_enable and q.bit == 8 and q.is_perchannel and q.is_symmetric
    q.dims = 4
    assert list(q._broadcast_qparams(torch.arange(5.0)).shape) == [5, 1, 1, 1]


def test_export_branch_stays_on_torch_builtins():
    """enable_export_onnx routes through torch.fake_quantize_* (quant_tensor.py:220-249): runs on CPU."""
    from sparsebit_amd.common import Backend
    from sparsebit_amd.config import quantizer_config
    from sparsebit_amd.quantizers import build_quantizer

    q = build_quantizer(quantizer_config("per-channel-symmetric", 8))
    q.set_backend(Backend.ONNXRUNTIME)
    q.dims = 2
    q.scale = q._broadcast_qparams(torch.tensor([0.1, 0.2, 0.05]))
    q.zero_point = q._broadcast_qparams(torch.zeros(3))
    q.enable_quant()
    q.enable_export_onnx()
    x = torch.tensor([[0.26, -1.0], [100.0, 0.31], [0.0, -0.024]])
    y = q(x)
    want = torch.fake_quantize_per_channel_affine(x, torch.tensor([0.1, 0.2, 0.05]), torch.zeros(3, dtype=torch.int32), 0, -128, 127)
    assert torch.equal(y, want)


def test_plugin_installs_into_reference_when_importable():
    """Only where /root/reference exists (authoring container): registry + native-module swap."""
    ref = "/root/reference"
    if not os.path.isdir(ref):
        pytest.skip("reference tree not present on this box")
    code = r'''
import sys, types, os
sys.path.insert(0, os.path.join(%r, "tests", "golden"))
import gen_golden
gen_golden.install_stubs()
sys.path.insert(0, %r)
import sparsebit.quantization.quantizers as rq, sparsebit.quantization.observers as ro
import sparsebit.quantization.quantizers.quant_tensor as rqt
sys.path.insert(0, %r)
import sparsebit_amd.plugin as plugin, sparsebit_amd.fake_quant as fq
info = plugin.install()
assert rqt.fake_quant_kernel is fq
assert rq.QUANTIZERS_MAP["uniform"].__module__.startswith("sparsebit_amd")
assert ro.OBSERVERS_MAP["percentile"].__module__.startswith("sparsebit_amd")
from sparsebit.quantization.common import QuantTarget
cfg = gen_golden.qcfg("per-channel-symmetric", 8)
q = rq.build_quantizer(cfg)
assert type(q).__module__ == "sparsebit_amd.quantizers.uniform", type(q)
for n in ("quant_pertensor_forward","quant_perchannel_forward","quant_pertensor_backward","quant_perchannel_backward"):
    assert callable(getattr(rqt.fake_quant_kernel, n))
print("PLUGIN_OK", info)
''' % (ROOT, ref, ROOT)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
    assert "PLUGIN_OK" in r.stdout, r.stdout


def test_gptq_pack_matches_reference_golden(golden):
    """QuantLinear.pack is host-side integer shuffling: runs anywhere."""
    from sparsebit_amd import gptq

    for name in ["g128", "g-1", "rag"] + golden["gptq23"].tolist():
        meta = [int(v) for v in golden["gptq/%s/meta" % name]]
        B, M, N, GS = meta[:4]
        bit = meta[4] if len(meta) > 4 else 4
        layer = torch.nn.Linear(M, N)
        with torch.no_grad():
            layer.weight.copy_(torch.from_numpy(golden["gptq/%s/wq" % name]))
            layer.bias.copy_(torch.from_numpy(golden["gptq/%s/bias" % name]))
        ql = gptq.QuantLinear(M, N, bit=bit, groupsize=GS)
        sh = (N, -1, 1) if GS != -1 and M // GS > 1 else (N, 1)
        ql.pack(layer, torch.from_numpy(golden["gptq/%s/scale" % name]).reshape(sh),
                torch.from_numpy(golden["gptq/%s/zero" % name]).reshape(sh))
        assert np.array_equal(ql.qweight.numpy(), golden["gptq/%s/qweight" % name])
        assert np.array_equal(ql.zeros.reshape(N, -1).numpy(), golden["gptq/%s/zeros" % name])
        assert sorted(ql.state_dict()) == ["bias", "qweight", "scales", "zeros"]


def test_pack32_to_pack8_matches_reference_script(golden):
    """convert_pack32topack8.py run on a tiny checkpoint (gen_golden.py) vs export.pack32_to_pack8"""
    from sparsebit_amd import export

    q32 = torch.from_numpy(golden["pack8/qweight32"])
    q8 = export.pack32_to_pack8(q32)
    assert q8.dtype == torch.int8
    assert np.array_equal(q8.numpy(), golden["pack8/qweight8"])
    assert torch.equal(export.pack8_to_pack32(q8), q32)
    sd = export.convert_checkpoint_pack32_to_pack8({"l.qweight": q32, "l.scales": torch.ones(2)})
    assert sd["l.qweight"].dtype == torch.int8 and sd["l.scales"].dtype == torch.float32
    with pytest.raises(TypeError):
        export.pack32_to_pack8(q32.to(torch.int64))


def test_int4_pack_roundtrip():
    from sparsebit_amd import export

    g = torch.Generator().manual_seed(0)
    lv = torch.randint(-8, 8, (6, 10), generator=g, dtype=torch.int8)
    p = export.pack_int4(lv)
    assert p.dtype == torch.uint8 and p.numel() == 30
    assert torch.equal(export.unpack_int4(p, True).reshape(6, 10), lv)
    lu = torch.randint(0, 16, (4, 8), generator=g, dtype=torch.uint8)
    assert torch.equal(export.unpack_int4(export.pack_int4(lu), False).reshape(4, 8), lu)
    assert int(export.pack_int4(torch.tensor([1, -1], dtype=torch.int8))[0]) == 0xF1  # element 0 low nibble


def test_group_table_build_is_host_side_and_validates():
    """sbq_group_table_build never touches the GPU: sizes, layout and error codes on the CPU"""
    import ctypes

    from sparsebit_amd import lib as L

    l = L.load()

    def item(C, inner, x=0x1000, y=0x2000, mask=None, qmin=-8, qmax=7):
        it = L.GroupItem()
        it.x, it.y, it.scale, it.zero_point, it.mask = x, y, 0x3000, 0x4000, mask
        it.C, it.inner, it.qmin, it.qmax, it.flags = C, inner, qmin, qmax, 0
        return it

    def build(items, buf=None):
        arr = (L.GroupItem * len(items))(*items)
        nt, need = ctypes.c_uint32(0), ctypes.c_size_t(0)
        rc = l.sbq_group_table_build(arr, len(items), buf.ctypes.data if buf is not None else None,
                                     buf.nbytes if buf is not None else 0, ctypes.byref(nt), ctypes.byref(need))
        return rc, nt.value, need.value

    items = [item(64, 576), item(1, 4096), item(3, 8)]  # 4608 + 512 + 3 packs -> 18 + 2 + 1 tiles
    rc, nt, need = build(items)
    assert rc == 0 and nt == 21 and need == 64 + 3 * 80 + 21 * 4
    buf = np.zeros(need, dtype=np.uint8)
    rc, nt2, _ = build(items, buf)
    assert rc == 0 and nt2 == 21
    words = buf.view(np.uint32)
    assert words[0] == 3 and words[1] == 21
    tile_item = words[(64 + 240) // 4:]
    assert tile_item.tolist() == [0] * 18 + [1] * 2 + [2]
    assert build(items, np.zeros(need - 4, dtype=np.uint8))[0] == 5  # SBQ_ERR_WORKSPACE
    assert build([item(8, 12)])[0] == 4  # inner % 8
    assert build([item(8, 16, x=0x1004)])[0] == 7  # SBQ_ERR_ALIGN
    assert build([item(8, 16), item(8, 16, mask=0x5000)])[0] == 4  # all masked or none
    assert build([item(8, 16, qmin=3, qmax=1)])[0] == 4
    assert build([item(0, 16)])[0] == 2  # empty
    assert build([item(1 << 20, 1 << 10)])[0] == 4  # >= 2^24 packs


def test_group_backward_table_build_on_host():
    import ctypes

    from sparsebit_amd import lib as L

    l = L.load()

    def item(C, inner, want_gs=1, mask=None, gx_off=0):
        it = L.GroupBwdItem()
        it.x, it.scale, it.zero_point, it.mask = 0x1000, 0x3000, 0x4000, mask
        it.gx_offset, it.gs_offset, it.C, it.inner = gx_off, 0, C, inner
        it.qmin, it.qmax, it.flags, it.want_gs, it.gs_ratio = -8, 7, L.GROUP_LSQ, want_gs, 0.5
        return it

    def build(items, buf=None):
        arr = (L.GroupBwdItem * len(items))(*items)
        wgs, rows, need, wsb = ctypes.c_uint32(0), ctypes.c_uint32(0), ctypes.c_size_t(0), ctypes.c_size_t(0)
        rc = l.sbq_group_bwd_table_build(arr, len(items), buf.ctypes.data if buf is not None else None,
                                         buf.nbytes if buf is not None else 0, ctypes.byref(wgs), ctypes.byref(rows),
                                         ctypes.byref(need), ctypes.byref(wsb))
        return rc, wgs.value, rows.value, need.value, wsb.value

    # 64 rows x 576 elements: 72 packs -> 2 segments per row -> 128 segments -> 32 workgroups;
    # 3 rows x 8 elements: 1 segment per row -> 3 segments -> 1 (padded) workgroup
    rc, wgs, rows, need, wsb = build([item(64, 576), item(3, 8)])
    assert (rc, wgs, rows) == (0, 33, 67)
    assert need == 64 + 2 * 96 + 33 * 4 and wsb >= 131 * 8
    buf = np.zeros(need, dtype=np.uint8)
    assert build([item(64, 576), item(3, 8)], buf)[0] == 0
    words = buf.view(np.uint32)
    assert words[:3].tolist() == [2, 33, 67]
    assert words[(64 + 192) // 4:].tolist() == [0] * 32 + [1]
    assert build([item(8, 12)])[0] == 4  # rows of whole packs only
    assert build([item(8, 16, gx_off=8)])[0] == 7  # gx offsets are multiples of 16
    assert build([item(8, 16), item(8, 16, mask=0x5000)])[0] == 4
    p = ctypes.c_void_p(0x1000)
    assert l.sbq_quant_group_backward(None, None, 1, 0, 0, 0, None, None, None, None, 0, None) == 3


def test_enums_compare_by_name_and_derived_classes_satisfy_foreign_isinstance():
    """The two mechanisms plugin.install() rests on, without the reference: (1) a foreign enum class with the
    reference's names compares / hashes equal to ours (modules/base.py:36-45 hands quantizers ITS members);
    (2) a class derived from (ours, foreign base) is an instance of the foreign base and never runs its __init__."""
    from enum import Enum

    import torch.nn as nn

    from sparsebit_amd import plugin
    from sparsebit_amd.common import Backend, QuantTarget
    from sparsebit_amd.config import quantizer_config
    from sparsebit_amd.quantizers import QUANTIZERS_MAP
    from sparsebit_amd.quantizers.quant_tensor import fake_quant_factory, fake_qrange_factory, ort_fake_quant, trt_fake_quant
    from sparsebit_amd.registry import impl_type

    ForeignBackend = Enum("Backend", {"VIRTUAL": 0, "ONNXRUNTIME": 1, "TENSORRT": 2})
    ForeignTarget = Enum("QuantTarget", {"WEIGHT": 0, "FEATURE": 1})
    Other = Enum("Other", {"VIRTUAL": 0})
    assert ForeignBackend.VIRTUAL == Backend.VIRTUAL and Backend.VIRTUAL == ForeignBackend.VIRTUAL
    assert not (ForeignBackend.VIRTUAL != Backend.VIRTUAL)
    assert ForeignBackend.TENSORRT != Backend.VIRTUAL and Backend.VIRTUAL != Other.VIRTUAL and Backend.VIRTUAL != 0
    assert fake_quant_factory[ForeignBackend.VIRTUAL] is ort_fake_quant
    assert fake_quant_factory[ForeignBackend.TENSORRT] is trt_fake_quant
    assert ForeignBackend.ONNXRUNTIME in fake_qrange_factory
    assert QuantTarget.FEATURE in (ForeignTarget.FEATURE,) and ForeignTarget.WEIGHT in [QuantTarget.WEIGHT]

    class ForeignBase(nn.Module):
        def __init__(self, config):
            raise AssertionError("foreign base __init__ ran")

        def only_on_foreign(self):
            return "kept"

    for name, cls in QUANTIZERS_MAP.items():
        d = plugin._derive(cls, ForeignBase)
        assert issubclass(d, cls) and issubclass(d, ForeignBase) and d.__name__ == cls.__name__
        assert plugin._derive(d, ForeignBase) is d
        target = "feature" if name == "pact" else "weight"
        cfg = quantizer_config("per-tensor-symmetric", 8, quantizer=name, target=target)
        cfg["TARGET"] = (ForeignTarget.FEATURE,) if name == "pact" else (ForeignTarget.WEIGHT,)
        q = d(cfg)  # PACT asserts FEATURE against the foreign member
        assert isinstance(q, ForeignBase) and impl_type(q) is cls and q.only_on_foreign() == "kept"
        q.set_backend(ForeignBackend.VIRTUAL)
        assert q.forward.__func__ is cls.forward  # behaviour is ours


def test_lsq_export_branch_builds_constants_without_a_host_round_trip():
    """LSQ's _qparams_preprocess under export_onnx (lsq.py:53-63): |scale| and clamp(zero_point) as graph-free
    tensors with the reference's values, and the export forward on torch.fake_quantize_*"""
    from sparsebit_amd.common import Backend
    from sparsebit_amd.config import quantizer_config
    from sparsebit_amd.quantizers import build_quantizer

    q = build_quantizer(quantizer_config("per-channel-symmetric", 4, "lsq"))
    q.set_backend(Backend.ONNXRUNTIME)
    q.dims = 2
    q.scale = torch.nn.Parameter(q._broadcast_qparams(torch.tensor([-0.1, 0.2, 0.05])))
    q.zero_point = q._broadcast_qparams(torch.tensor([0.0, 9.0, -20.0]))
    q.init_params = True
    q.enable_quant()
    q.enable_export_onnx()
    s, z = q._qparams_preprocess(None)
    assert not s.requires_grad and s.grad_fn is None and z.grad_fn is None
    assert torch.equal(s.reshape(-1), torch.tensor([0.1, 0.2, 0.05])) and torch.equal(z.reshape(-1), torch.tensor([0.0, 7.0, -8.0]))
    assert s.data_ptr() != q.scale.data_ptr()  # a copy: the exporter may not alias the Parameter
    x = torch.tensor([[0.26, -1.0], [100.0, 0.31], [0.0, -0.024]])
    y = q(x)
    want = torch.fake_quantize_per_channel_affine(x, torch.tensor([0.1, 0.2, 0.05]), torch.tensor([0, 7, -8], dtype=torch.int32), 0, -128, 127)
    assert y.shape == want.shape


def test_plugin_preinstall_answers_the_reference_import_time_jit():
    """plugin.preinstall(): the reference's `load(name="fake_quant", ...)` at import (quant_tensor.py:7-22) gets the
    prebuilt HIP module instead of a hipify + compile of its CUDA sources; other extensions still go to torch"""
    import torch.utils.cpp_extension as ext

    from sparsebit_amd import fake_quant, plugin

    real = ext.load
    restore = plugin.preinstall()
    try:
        assert ext.load is not real
        got = ext.load(name="fake_quant", sources=["/nonexistent/export.cc"], with_cuda=True, build_directory="/nonexistent")
        assert got is fake_quant
        for fn in ("quant_pertensor_forward", "quant_perchannel_forward", "quant_pertensor_backward", "quant_perchannel_backward"):
            assert hasattr(got, fn)
        assert plugin.preinstall()() is None  # idempotent: a second call wraps nothing
        with pytest.raises(Exception):
            ext.load(name="something_else", sources=["/nonexistent/x.cc"], build_directory="/nonexistent")
    finally:
        restore()
    assert ext.load is real


# ---- round 5: launch plans / captured graphs -- the host-side bookkeeping (no kernel runs here) -----------------------
def test_structure_version_and_epoch_follow_every_switch():
    """a launch plan (sparsebit_amd.plan) is valid for ONE structure version of its quantizer, a captured graph
    (sparsebit_amd.graph) for ONE process-wide epoch: both must move with everything that changes what a forward does"""
    import copy
    import io

    from sparsebit_amd import plan
    from sparsebit_amd.common import Backend
    from sparsebit_amd.config import quantizer_config
    from sparsebit_amd.quantizers import build_quantizer
    from sparsebit_amd.quantizers import quant_tensor as QT

    q = build_quantizer(quantizer_config("per-channel-symmetric", 8))
    for change in (lambda: q.enable_quant(), lambda: q.disable_quant(), lambda: q.set_backend(Backend.TENSORRT),
                   lambda: q.enable_export_onnx(), lambda: q.disable_export_onnx(), lambda: setattr(q, "keep_input_dtype", True),
                   lambda: setattr(q, "scale", torch.ones(3)), lambda: setattr(q, "zero_point", torch.zeros(3)),
                   lambda: q.set_fake_fused(), lambda: q.float()):
        sv, ep = q._sv, plan.epoch()
        change()
        assert q._sv > sv and plan.epoch() > ep
    qv, ep = q.qdesc.version, plan.epoch()
    q.set_bit(4)
    assert q.qdesc.version > qv and plan.epoch() > ep and q.qdesc.qrange == (-8, 7)
    qv = q.qdesc.version
    q.qdesc.set_symmetric(False)  # straight on the descriptor (lsq.py:39-43 does)
    assert q.qdesc.version > qv
    # the process default of the output dtype, and a quantizer's own choice
    q2 = build_quantizer(quantizer_config("per-tensor-affine", 8))
    assert q2.keep_input_dtype is None and q2._out_keeps_dtype() is False
    ep = plan.epoch()
    QT.keep_input_dtype(True)
    try:
        assert q2._out_keeps_dtype() is True and plan.epoch() > ep
        q2.keep_input_dtype = False
        assert q2._out_keeps_dtype() is False and q2._out_dtype(torch.zeros(1, dtype=torch.bfloat16)) == torch.float32
    finally:
        QT.keep_input_dtype(False)
    # copies start without a plan (a plan holds a foreign function and device addresses)
    q3 = copy.deepcopy(q2)
    assert q3._plans is not q2._plans and q3._plans.plan is None
    buf = io.BytesIO()
    torch.save(q2, buf)
    buf.seek(0)
    q4 = torch.load(buf, weights_only=False)
    assert q4._plans.plan is None and q4.keep_input_dtype is False
    sd = q2.state_dict()
    assert sorted(k for k in sd if "observer" not in k) == ["scale", "zero_point"]  # nothing new in checkpoints


def test_graph_module_refuses_host_tensors_and_training_mode():
    from sparsebit_amd import graph

    m = torch.nn.Linear(4, 4)
    with pytest.raises(RuntimeError):
        graph.capture(m, torch.zeros(2, 4))  # training mode
    with pytest.raises(RuntimeError):
        graph.capture(m.eval(), torch.zeros(2, 4))  # host tensors
    with pytest.raises(ValueError):
        graph.capture(m, torch.zeros(2, 4), on_stale="ignore")
