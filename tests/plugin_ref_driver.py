"""Drives the REAL reference (imported from /root/reference, authoring container only) with
sparsebit_amd.plugin installed, on a box without a GPU, and prints one JSON line of findings.
Run by tests/test_plugin_reference.py in a subprocess with the GPUs hidden.

What it proves (VERDICT r01 "What's missing" 1): quantizers BUILT BY THE REFERENCE
(`build_quantizer(cfg)` with the reference's TARGET / Backend enum members, modules/base.py:36-45)
run their forward up to the point where the HIP library asks for a device tensor -- i.e. no
KeyError on the backend, no "PACT only support feature quantization" -- and QuantModel.export_onnx's
isinstance-driven enable loop (quant_model.py:236,256) reaches every installed quantizer and the
export branch computes with torch builtins on the CPU.
"""
import json
import os
import sys
import tempfile

os.environ["HIP_VISIBLE_DEVICES"] = ""
os.environ["CUDA_VISIBLE_DEVICES"] = ""
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, ROOT)
import gen_golden  # noqa: E402

gen_golden.install_stubs()
sys.path.insert(0, "/root/reference")

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

import sparsebit.quantization.quantizers as rq  # noqa: E402
import sparsebit.quantization.observers as ro  # noqa: E402
from sparsebit.quantization import QuantModel, parse_qconfig  # noqa: E402
from sparsebit.quantization.common import Backend as RefBackend, QuantTarget as RefTarget  # noqa: E402

import sparsebit_amd.plugin as plugin  # noqa: E402
from sparsebit_amd.lib import SbqError  # noqa: E402

out = {}
calibrate = "device" if "--device-calibration" in sys.argv else None
out["installed"] = plugin.install(calibrate=calibrate)


def stops_at_device_check(fn):
    """'device' when fn dies in lib.require_device (the furthest a box without a GPU can get)."""
    try:
        fn()
    except SbqError as e:
        return "device" if "MI355X only" in str(e) else "SbqError: " + str(e)
    except Exception as e:  # KeyError on the backend, PACT's target assertion, ...
        return "{}: {}".format(type(e).__name__, e)
    return "returned"


# ---- (i) quantizers built through the reference, reference enums ------------------------------------
cases = [
    ("uniform", "MINMAX", True, "per-channel-symmetric"),
    ("uniform", "MSE", False, "per-tensor-affine"),
    ("uniform", "PERCENTILE", False, "per-tensor-symmetric"),
    ("uniform", "ACIQ", False, "per-tensor-symmetric"),
    ("uniform", "MOVING_AVERAGE", False, "per-tensor-affine"),
    ("lsq", "MINMAX", True, "per-channel-symmetric"),
    ("lsq+", "MINMAX", False, "per-tensor-affine"),
    ("pact", "MINMAX", False, "per-tensor-symmetric"),
    ("dorefa", "MINMAX", True, "per-tensor-symmetric"),
]
built = {}
for qtype, obs, is_w, scheme in cases:
    key = "{}/{}/{}".format(qtype, obs, "W" if is_w else "A")
    rec = {}
    try:
        cfg = gen_golden.qcfg(scheme, 8, observer=obs, quantizer=qtype, target_weight=is_w)
        assert type(cfg.TARGET[0]) is RefTarget
        q = rq.build_quantizer(cfg)
        q.set_backend(RefBackend.VIRTUAL)
        rec["module"] = type(q).__module__
        rec["isinstance_ref_quantizer"] = isinstance(q, rq.Quantizer)
        rec["isinstance_ref_observer"] = isinstance(q.observer, ro.Observer)
        rec["target_is_feature"] = bool(q.qdesc.target == RefTarget.FEATURE)
        x = torch.randn(4, 6, 5, 5) if not is_w else torch.randn(6, 20)
        rec["update_observer"] = stops_at_device_check(lambda: q.update_observer(x))  # DoReFa transforms on the device
        if rec["update_observer"] == "returned":
            rec["calc_qparams"] = stops_at_device_check(q.calc_qparams)
        # forward with hand-set qparams: reaches fake_quant_factory[ref backend] -> ops -> require_device
        q.observer.data_cache.reset()
        q.dims = x.dim()
        C = x.shape[q.qdesc.ch_axis] if q.is_perchannel else 1
        if isinstance(q.scale, nn.Parameter) or qtype in ("lsq", "lsq+"):
            q.scale = nn.Parameter(q._broadcast_qparams(torch.full((C,), 0.05)))
            q.zero_point = nn.Parameter(q._broadcast_qparams(torch.zeros(C))) if qtype == "lsq+" else q._broadcast_qparams(torch.zeros(C))
            q.init_params = True
        else:
            q.scale = q._broadcast_qparams(torch.full((C,), 0.05))
            q.zero_point = q._broadcast_qparams(torch.zeros(C))
        if qtype == "pact":
            q.alpha = nn.Parameter(torch.tensor([3.0]))
        q.enable_quant()
        for backend in (RefBackend.VIRTUAL, RefBackend.ONNXRUNTIME, RefBackend.TENSORRT):
            q.set_backend(backend)
            rec["forward/" + backend.name] = stops_at_device_check(lambda: q(x))
    except Exception as e:
        rec["error"] = "{}: {}".format(type(e).__name__, e)
    built[key] = rec
out["built"] = built

# ---- (ii) a reference QuantModel: build, BN fusion, export_onnx's enable loop -------------------------
cfg_text = """
BACKEND: virtual
SCHEDULE:
  FUSE_BN: True
W:
  QSCHEME: per-channel-symmetric
  QUANTIZER:
    TYPE: uniform
    BIT: 8
  OBSERVER:
    TYPE: MINMAX
A:
  QSCHEME: per-tensor-affine
  QUANTIZER:
    TYPE: uniform
    BIT: 8
  OBSERVER:
    TYPE: MINMAX
    LAYOUT: NCHW
"""


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.c1 = nn.Conv2d(3, 8, 3, padding=1)
        self.bn = nn.BatchNorm2d(8)
        self.r = nn.ReLU()
        self.c2 = nn.Conv2d(8, 16, 3, padding=1)
        self.p = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(16, 10)

    def forward(self, x):
        x = self.r(self.bn(self.c1(x)))
        x = self.r(self.c2(x))
        return self.fc(torch.flatten(self.p(x), 1))


with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
    f.write(cfg_text)
torch.manual_seed(0)
qm = QuantModel(Net().eval(), parse_qconfig(f.name))
os.unlink(f.name)
quantizers = [(n, m) for n, m in qm.model.named_modules() if isinstance(m, rq.Quantizer)]
out["qmodel_quantizers"] = len(quantizers)
out["qmodel_all_amd"] = all(type(m).__module__.startswith("sparsebit_amd") for _, m in quantizers)
out["qmodel_fused"] = sorted(n for n, m in quantizers if m.fake_fused)

# calibration: the reference's (or, with --device-calibration, the routed) runner reaches the device check
qm.prepare_calibration()
out["calibration_runner"] = type(qm.calibration_runner).__module__
rec = stops_at_device_check(lambda: [qm(torch.randn(2, 3, 8, 8)), qm.calc_qparams()])
out["calibration"] = rec
if hasattr(getattr(qm, "calibration_runner", None), "abort"):  # the routed runner died inside its hook: unhook
    qm.calibration_runner.abort()

# qparams by hand (what calibration would have produced), then the export loop
g = torch.Generator().manual_seed(1)
for n, m in quantizers:
    if m.fake_fused:
        continue
    is_w = n.endswith("weight_quantizer")
    if is_w:
        w = qm.model.get_submodule(n.rsplit(".", 1)[0]).weight
        m.dims = w.dim()
        C = w.shape[0]
        m.scale = m._broadcast_qparams(torch.rand(C, generator=g) * 0.02 + 0.005)
        m.zero_point = m._broadcast_qparams(torch.zeros(C))
    else:
        m.dims = 4 if "fc" not in n else 2
        m.scale = m._broadcast_qparams(torch.tensor([0.03]))
        m.zero_point = m._broadcast_qparams(torch.tensor([17.0]))

seen = {}
dummy = torch.randn(2, 3, 8, 8, generator=g)
import torch.onnx  # noqa: E402


def fake_export(model, data, name, **kw):
    """stand-in for torch.onnx.export (its exporter needs `onnx`, absent from this image): what tracing
    does to the quantizers -- one forward on the CPU -- with the flags recorded."""
    seen["export_flags"] = [m.export_onnx for _, m in quantizers]
    seen["devices"] = sorted({str(p.device) for p in model.parameters()})
    with torch.no_grad():
        seen["y"] = model(data)


real_export = torch.onnx.export
torch.onnx.export = fake_export
try:
    qm.export_onnx(dummy, "/tmp/unused.onnx")
finally:
    torch.onnx.export = real_export
out["export_all_enabled"] = all(seen["export_flags"]) and len(seen["export_flags"]) == len(quantizers)
out["export_devices"] = seen["devices"]
out["export_flags_after"] = [m.export_onnx for _, m in quantizers]

# the same forward with torch's own fake-quant ops placed by hand == what the export branch computed
with torch.no_grad():
    def fq_a(x, m):
        return torch.fake_quantize_per_tensor_affine(x, m.scale.item(), int(m.zero_point.item()), 0, 255)

    def fq_w(w, m):
        return torch.fake_quantize_per_channel_affine(w, m.scale.reshape(-1), m.zero_point.reshape(-1).int(), 0, -128, 127)

    mm = dict(qm.model.named_modules())
    c1, c2, fc = mm["c1_bn"], mm["c2"], mm["fc"]
    x = torch.nn.functional.conv2d(fq_a(dummy, c1.input_quantizer), fq_w(c1.weight, c1.weight_quantizer), c1.bias, padding=1)
    x = torch.relu(x)
    x = torch.nn.functional.conv2d(fq_a(x, c2.input_quantizer), fq_w(c2.weight, c2.weight_quantizer), c2.bias, padding=1)
    x = torch.relu(x)
    pq = mm["p"].input_quantizer
    x = torch.nn.functional.adaptive_avg_pool2d(fq_a(x, pq) if not pq.fake_fused else x, 1)
    x = torch.flatten(x, 1)
    want = torch.nn.functional.linear(fq_a(x, fc.input_quantizer), fq_w(fc.weight, fc.weight_quantizer), fc.bias)
out["export_matches_torch_builtins"] = bool(torch.equal(seen["y"], want))


# ---- the sparse side: SparseModel built by the reference, calc_params routed model-wide (round 6) ---------------------
try:
    from sparsebit.sparse import SparseModel, parse_sconfig

    with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
        f.write("SPARSER:\n  TYPE: unstructed\n  STRATEGY: l1norm\n  RATIO: 0.5\n")
    import contextlib
    import io

    with contextlib.redirect_stdout(io.StringIO()):  # (the reference prints the traced graph)
        sm = SparseModel(Net().eval(), parse_sconfig(f.name))
    os.unlink(f.name)
    sparsers = [m.sparser for m in sm.model.modules() if getattr(m, "sparser", None) is not None]
    out["sparse_sparsers"] = len(sparsers)
    out["sparse_all_amd"] = all(type(sp).__module__.startswith("sparsebit_amd") for sp in sparsers)
    out["sparse_calc_params_routed"] = bool(getattr(SparseModel.calc_params, "_sbq_grouped", False))
    # CPU weights: nothing is grouped, the reference's loop runs and the first layer's sparser reaches the device check
    out["sparse_calc_params"] = stops_at_device_check(sm.calc_params)
    out["sparse_premask_left"] = any(getattr(sp, "_premask", None) is not None for sp in sparsers)
except Exception as e:  # noqa: BLE001
    out["sparse_error"] = "{}: {}".format(type(e).__name__, e)
print("PLUGIN_JSON " + json.dumps(out))
