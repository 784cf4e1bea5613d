"""The reference-side drop-in on the GPU box, where /root/reference does not exist.

A reference-SHAPED harness stands in for the reference: its own `Backend` / `QuantTarget` enum classes
(sparsebit/quantization/common.py:5-35), its own `Quantizer` / `Observer` base classes, a QuantOpr
look-alike whose `build_quantizer` sets TARGET and the backend with THOSE enums
(modules/base.py:36-45), and a model whose `export_onnx` finds quantizers with `isinstance(m, Quantizer)`
against THAT base (quant_model.py:236,256).  The classes handed to it come from the same derivation
plugin.install() performs (`plugin._derive`).  tests/test_plugin_reference.py runs the real reference
through the same mechanics on the CPU container; here the kernels actually execute.

The numbers: tests/golden/calib_golden.npz holds what the reference's own CalibrationRunner computed on the
CPU for this operator chain (tests/golden/gen_calib_golden.py), in both asym modes.
"""
import os
from enum import Enum

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from conftest import ROOT
from helpers import same_values

pytestmark = pytest.mark.gpu


# ---- the stand-in "reference" -----------------------------------------------------------------------
class Backend(Enum):  # a DIFFERENT class object than sparsebit_amd.common.Backend, same names
    VIRTUAL = 0
    ONNXRUNTIME = 1
    TENSORRT = 2


class QuantTarget(Enum):
    WEIGHT = 0
    FEATURE = 1


class Quantizer(nn.Module):  # what `from sparsebit.quantization.quantizers import Quantizer` gives quant_model.py
    def __init__(self, config):
        raise AssertionError("the reference base's __init__ must not run for an installed class")


class Observer(nn.Module):
    def __init__(self, config, qdesc):
        raise AssertionError("the reference base's __init__ must not run for an installed class")


def _registries():
    from sparsebit_amd import observers as amd_o
    from sparsebit_amd import plugin
    from sparsebit_amd import quantizers as amd_q

    qmap = {k: plugin._derive(getattr(v, "_sbq_impl", v), Quantizer) for k, v in amd_q.QUANTIZERS_MAP.items()}
    return qmap, amd_o


def build_quantizer(cfg):
    qmap, _ = _registries()
    return qmap[cfg.QUANTIZER.TYPE.lower()](cfg)


class QuantOpr(nn.Module):
    def __init__(self):
        super().__init__()
        self.weight = None
        self.input_quantizer = None
        self.weight_quantizer = None

    def build_quantizer(self, wcfg, acfg, backend):
        if self.weight is not None:
            wcfg["TARGET"] = (QuantTarget.WEIGHT,)
            self.weight_quantizer = build_quantizer(wcfg)
            self.weight_quantizer.set_backend(backend)
        acfg["TARGET"] = (QuantTarget.FEATURE,)
        self.input_quantizer = build_quantizer(acfg)
        self.input_quantizer.set_backend(backend)

    def set_quant(self, w_quant=False, a_quant=False):
        for q, on in ((self.weight_quantizer, w_quant), (self.input_quantizer, a_quant)):
            if q:
                if on and not q.fake_fused:
                    q.enable_quant()
                else:
                    q.disable_quant()


class QConv2d(QuantOpr):  # modules/conv.py:30-36
    def __init__(self, w, b):
        super().__init__()
        self.weight, self.bias = nn.Parameter(w), nn.Parameter(b)

    def forward(self, x):
        return F.conv2d(self.input_quantizer(x), self.weight_quantizer(self.weight), self.bias, padding=1)


class QLinear(QuantOpr):  # modules/linear.py:30-34
    def __init__(self, w, b):
        super().__init__()
        self.weight, self.bias = nn.Parameter(w), nn.Parameter(b)

    def forward(self, x):
        return F.linear(self.input_quantizer(x), self.weight_quantizer(self.weight), self.bias)


class QReLU(QuantOpr):  # modules/activations.py: input quantizer fused away by DISABLE_UNNECESSARY_QUANT
    def forward(self, x):
        return F.relu(self.input_quantizer(x))


class QPool(QuantOpr):
    def forward(self, x):
        return F.adaptive_avg_pool2d(self.input_quantizer(x), 1)


class Chain(nn.Module):
    """c1_bn -> r -> c2 -> r_1 -> p -> flatten -> fc: the graph QuantModel makes of gen_calib_golden.Net"""

    def __init__(self, z, name, wcfg, acfg, backend):
        super().__init__()
        t = lambda k: torch.from_numpy(z["{}/{}".format(name, k)].copy())
        self.c1_bn = QConv2d(t("c1_bn/weight"), t("c1_bn/bias"))
        self.r = QReLU()
        self.c2 = QConv2d(t("c2/weight"), t("c2/bias"))
        self.r_1 = QReLU()
        self.p = QPool()
        self.fc = QLinear(t("fc/weight"), t("fc/bias"))
        for m in (self.c1_bn, self.r, self.c2, self.r_1, self.p, self.fc):
            m.build_quantizer(wcfg(), acfg(), backend)

    def forward(self, x):
        x = self.r(self.c1_bn(x))
        x = self.r_1(self.c2(x))
        return self.fc(torch.flatten(self.p(x), 1))

    def export_onnx(self, dummy, trace):  # quant_model.py:222-258 with the exporter passed in
        self.eval()
        for m in self.modules():
            if isinstance(m, QuantOpr):
                m.set_quant(True, True)
        hit = 0
        for m in self.modules():
            if isinstance(m, Quantizer):
                m.enable_export_onnx()
                hit += 1
        y = trace(self.cpu(), dummy)
        for m in self.modules():
            if isinstance(m, Quantizer):
                m.disable_export_onnx()
        return hit, y


@pytest.fixture(scope="module")
def calib():
    return np.load(os.path.join(ROOT, "tests", "golden", "calib_golden.npz"), allow_pickle=False)


_SCHEMES = {
    "minmax": ("per-channel-symmetric", "MINMAX", "per-tensor-affine", "MINMAX"),
    "pct_mse": ("per-channel-symmetric", "PERCENTILE", "per-tensor-symmetric", "MSE"),
}


def _chain(calib, name):
    from sparsebit_amd.config import quantizer_config

    ws, wo, as_, ao = _SCHEMES[name]
    wcfg = lambda: quantizer_config(ws, 8, observer=wo)
    acfg = lambda: quantizer_config(as_, 8, observer=ao, target="feature", layout="NCHW")
    model = Chain(calib, name, wcfg, acfg, Backend.VIRTUAL)
    fused = [n for n in calib[name + "/asym0/names"].tolist() if int(calib["{}/asym0/{}/fake_fused".format(name, n)])]
    for n in fused:  # what the reference's DISABLE_UNNECESSARY_QUANT pass did to its graph
        model.get_submodule(n).set_fake_fused()
    return model.cuda(), fused


def test_foreign_enums_and_bases_drive_the_hip_path(calib):
    from sparsebit_amd.common import Backend as AmdBackend
    from sparsebit_amd.common import QuantTarget as AmdTarget

    assert Backend is not AmdBackend and Backend.VIRTUAL == AmdBackend.VIRTUAL and AmdTarget.FEATURE == QuantTarget.FEATURE
    assert Backend.TENSORRT != AmdBackend.VIRTUAL
    model, fused = _chain(calib, "minmax")
    assert fused == ["r.input_quantizer", "r_1.input_quantizer"]
    qs = [m for m in model.modules() if isinstance(m, Quantizer)]  # the reference's isinstance check
    assert len(qs) == 9 and all(type(q).__module__.startswith("sparsebit_amd.quantizers") for q in qs)
    assert type(model.c2.input_quantizer.qdesc.target) is QuantTarget
    # PACT accepts the foreign FEATURE member ("PACT only support feature quantization" raised)
    from sparsebit_amd.config import quantizer_config

    cfg = quantizer_config("per-tensor-symmetric", 8, quantizer="pact", target="feature")
    cfg["TARGET"] = (QuantTarget.FEATURE,)
    pact = build_quantizer(cfg)
    pact.set_backend(Backend.ONNXRUNTIME)
    x = torch.randn(2, 4, 5, 5, device="cuda")
    pact.update_observer(x)
    pact.calc_qparams()
    pact.enable_quant()
    assert pact(x).shape == x.shape
    # TensorRT member of the foreign enum picks the symmetric-only route (zero point assertion)
    q = model.c1_bn.input_quantizer
    q.update_observer(x[:, :3])
    q.calc_qparams()
    q.enable_quant()
    q.set_backend(Backend.TENSORRT)
    if q.zero_point.abs().sum() != 0:
        with pytest.raises(AssertionError, match="tensorrt only support symmetric"):
            q(x[:, :3])


@pytest.mark.parametrize("name", ["minmax", "pct_mse"])
@pytest.mark.parametrize("asym", [False, True])
def test_device_calibrator_equals_reference_calibration_runner(calib, name, asym):
    """prepare_calibration -> the user's forwards -> calc_qparams(asym): every scale / zero_point the
    reference's CalibrationRunner produced on the CPU (both modes; they agree, see gen_calib_golden.py)."""
    from sparsebit_amd.calibration import DeviceCalibrator

    model, fused = _chain(calib, name)
    runner = DeviceCalibrator(model)
    runner.prepare_calibration()
    with torch.no_grad():
        for b in calib["batches"]:
            model(torch.from_numpy(b).cuda())
    res = runner.layerwise_calibration(torch.device("cuda"), asym, asym, asym)
    tag = "{}/asym{}".format(name, int(asym))
    names = calib[tag + "/names"].tolist()
    live = [n for n in names if n not in fused]
    assert sorted(res) == sorted(live)
    for n in names:
        q = model.get_submodule(n)
        gs, gz = calib["{}/{}/scale".format(tag, n)], calib["{}/{}/zero_point".format(tag, n)]
        s, z = q.scale.reshape(-1).cpu().numpy(), q.zero_point.reshape(-1).cpu().numpy()
        if n.endswith("weight_quantizer") or n == "c1_bn.input_quantizer":
            # fed by the stored weights / the raw calibration batches: identical data on both sides -> bit-exact
            assert np.array_equal(s, gs), (n, s, gs)
            assert same_values(z, gz), (n, z, gz)
        else:
            # fed by an activation a MIOpen / rocBLAS operator produced here and the CPU's conv produced in the
            # golden run: the observed tensors differ in the last ulp (summation order of the float operator, not
            # of this path), so min/max and hence the scale may move by an ulp, a zero point by one level
            assert np.allclose(s, gs, rtol=1e-5, atol=0), (n, s, gs)
            assert np.abs(z - gz).max() <= 1, (n, z, gz)
    # quantized end-to-end forward of the calibrated chain vs the reference's CPU result: same grid everywhere,
    # MIOpen vs CPU conv summation order may move an activation across a rounding boundary (one level = scale)
    if not asym:
        for m in model.modules():
            if isinstance(m, QuantOpr):
                m.set_quant(True, True)
        with torch.no_grad():
            y = model(torch.from_numpy(calib["batches"][0]).cuda()).cpu().numpy()
        ref, flt = calib[name + "/y_quant"], calib[name + "/y_float"]
        assert np.abs(y - ref).max() <= 0.25 * np.abs(ref - flt).max() + 1e-3, (np.abs(y - ref).max(), np.abs(ref - flt).max())


def test_export_loop_finds_installed_quantizers(calib):
    """quant_model.py:236-258: enable by isinstance, trace on the CPU with torch builtins, disable."""
    model, _ = _chain(calib, "minmax")
    from sparsebit_amd.calibration import DeviceCalibrator

    DeviceCalibrator(model).calibrate([torch.from_numpy(b).cuda() for b in calib["batches"]])
    dummy = torch.from_numpy(calib["batches"][0])
    flags = []

    def trace(cpu_model, data):
        flags.extend(m.export_onnx for m in cpu_model.modules() if isinstance(m, Quantizer))
        with torch.no_grad():
            return cpu_model(data)

    hit, y_export = model.export_onnx(dummy, trace)
    assert hit == 9 and all(flags)
    assert not any(m.export_onnx for m in model.modules() if isinstance(m, Quantizer))
    # the export branch (torch.fake_quantize_*, CPU) and the HIP path agree on the calibrated model
    model.cuda()
    with torch.no_grad():
        y_hip = model(dummy.cuda()).cpu()
    assert (y_export - y_hip).abs().max() <= 0.05 * y_hip.abs().max()
