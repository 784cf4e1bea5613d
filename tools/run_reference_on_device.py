"""Run the REAL reference -- megvii-research/Sparsebit's QuantModel, built from its own example model
(examples/quantization_aware_training/cifar10/basecase/model.py: resnet20) -- on the MI355X with
sparsebit_amd.plugin.install(calibrate="device"), and compare it with the same model run by the reference ALONE on the
host (its CPU path, GPUs hidden, no plugin) in a subprocess of this very script.

    python tools/run_reference_on_device.py [--reference /path/to/Sparsebit] > profiles/r04_reference_qmodel_on_device.log

The reference tree is not part of this repository and must not be copied into it; a maintainer points --reference (or
$SBQ_REFERENCE) at a checkout.  (For the round-4 run on the GPU box the tree travelled as an untracked, git-ignored
directory `_reference_tmp/` that was deleted afterwards.)  The three imports the reference needs and this image lacks
(yacs, onnx, torchvision.ops.stochastic_depth) are stubbed exactly as for the golden generators (tests/golden/gen_golden.py).

What is run, in both processes from the same seeds:
  PTQ   QuantModel(resnet20), W per-channel-symmetric int8 / A per-tensor-affine uint8, min-max observers, BN fusion:
        prepare_calibration -> 4 calibration forwards -> calc_qparams -> set_quant -> quantized forward
  QAT   qconfig_lsq.yaml of that example (LSQ 4w4a, first / last layer 8 bit as main.py:155-158): init_QAT -> forward ->
        cross-entropy -> backward -> one SGD step (device only: the reference's CPU path has no STE backward,
        quant_tensor.py:113-116)
  export the reference's own QuantModel.export_onnx loop with torch.onnx.export replaced by a forward (its exporter
        needs `onnx`), and sparsebit_amd.export.save_qdq_onnx of the same model (hand-written protobuf)
Compared: every quantizer's scale / zero_point (the device convolutions are MIOpen's, the host's are oneDNN's: activations
differ in the last bits, so scales agree to ~1e-6 relative, not bit for bit; weights' scales are bit-exact) and the logits.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

PTQ_YAML = """
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


def find_reference(arg):
    for p in (arg, os.environ.get("SBQ_REFERENCE"), os.path.join(ROOT, "_reference_tmp"), "/root/reference"):
        if p and os.path.isdir(os.path.join(p, "sparsebit")):
            return os.path.abspath(p)
    raise SystemExit("no reference checkout found: pass --reference /path/to/Sparsebit (the directory that holds sparsebit/)")


def setup(ref):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import gen_golden

    gen_golden.install_stubs()
    import importlib.machinery

    for name in ("onnx", "yacs", "yacs.config", "torchvision", "torchvision.ops", "torchvision.ops.stochastic_depth"):
        # (torch._dynamo, imported by torch.optim, walks sys.modules with importlib.util.find_spec: a stub needs a spec)
        if getattr(sys.modules[name], "__spec__", None) is None:
            sys.modules[name].__spec__ = importlib.machinery.ModuleSpec(name, None)
    sys.path.insert(0, ref)
    sys.path.insert(0, os.path.join(ref, "examples", "quantization_aware_training", "cifar10", "basecase"))


def build(ref, yaml_text, device):
    import torch
    from model import resnet20
    from sparsebit.quantization import QuantModel, parse_qconfig

    with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
        f.write(yaml_text)
    torch.manual_seed(0)
    net = resnet20(num_classes=10)
    # non-trivial BN statistics, so that BN fusion has something to fold
    g = torch.Generator().manual_seed(3)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.weight.data.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)
    qm = QuantModel(net.eval(), parse_qconfig(f.name))
    os.unlink(f.name)
    return qm.to(device) if device != "cpu" else qm


def batches(n, device):
    import torch

    g = torch.Generator().manual_seed(11)
    return [torch.randn(16, 3, 32, 32, generator=g).to(device) for _ in range(n)]


def qparams(qm):
    import sparsebit.quantization.quantizers as rq

    out = {}
    for n, m in qm.model.named_modules():
        if isinstance(m, rq.Quantizer) and not m.fake_fused:
            out[n] = (m.scale.detach().reshape(-1).float().cpu(), m.zero_point.detach().reshape(-1).float().cpu())
    return out


def ptq(qm, device):
    import torch

    qm.prepare_calibration()
    with torch.no_grad():
        for b in batches(4, device):
            qm(b)
    qm.calc_qparams()
    qm.set_quant(w_quant=True, a_quant=True)
    with torch.no_grad():
        y = qm(batches(5, device)[4])
    return qparams(qm), y.float().cpu()


def host_leg(ref, out_path):
    """the reference alone, CPU path"""
    import torch

    assert not torch.cuda.is_available()
    setup(ref)
    qm = build(ref, PTQ_YAML, "cpu")
    t = time.perf_counter()
    qp, y = ptq(qm, "cpu")
    dt = time.perf_counter() - t
    qm2 = build(ref, open(os.path.join(ref, "examples", "quantization_aware_training", "cifar10", "basecase", "qconfig_lsq.yaml")).read(), "cpu")
    for name in ("conv1", "fc"):
        getattr(qm2.model, name).input_quantizer.set_bit(bit=8)
        getattr(qm2.model, name).weight_quantizer.set_bit(bit=8)
    qm2.prepare_calibration()
    with torch.no_grad():
        for b in batches(4, "cpu"):
            qm2(b)
        qm2.init_QAT()
    torch.save({"ptq_qparams": qp, "ptq_logits": y, "ptq_seconds": dt, "lsq_qparams": qparams(qm2)}, out_path)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default=None)
    ap.add_argument("--host-leg", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    ref = find_reference(args.reference)
    if args.host_leg:
        return host_leg(ref, args.host_leg)

    import torch

    assert torch.cuda.is_available(), "the device leg needs the MI355X"
    print("reference:", ref)
    # ---- the reference alone on the host ----
    tmp = tempfile.mkdtemp()
    host_pt = os.path.join(tmp, "host.pt")
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--reference", ref, "--host-leg", host_pt], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-4000:]
    host = torch.load(host_pt)
    print("host leg (reference alone, CPU path): PTQ calibration + forward of QuantModel(resnet20) in %.2f s, %d live quantizers"
          % (host["ptq_seconds"], len(host["ptq_qparams"])))

    # ---- the same model on the device under the plugin ----
    sys.path.insert(0, ROOT)
    import sparsebit_amd.plugin as plugin

    plugin.preinstall()  # the reference JIT-builds its CUDA extension at import as soon as a GPU is visible
    setup(ref)
    import sparsebit.quantization.quantizers as rq

    info = plugin.install(calibrate="device")
    print("plugin.install:", json.dumps(info))
    dev = "cuda"
    qm = build(ref, PTQ_YAML, dev)
    quantizers = [(n, m) for n, m in qm.model.named_modules() if isinstance(m, rq.Quantizer)]
    print("QuantModel(resnet20): %d quantizers, all sparsebit_amd classes: %s, calibration runner: %s" % (
        len(quantizers), all(type(m).__module__.startswith("sparsebit_amd") for _, m in quantizers), "DeviceCalibrator"))
    torch.cuda.synchronize()
    t = time.perf_counter()
    qp, y = ptq(qm, dev)
    torch.cuda.synchronize()
    print("device leg: PTQ calibration + forward in %.3f s (first call: includes library load)" % (time.perf_counter() - t))
    worst_s = worst_z = 0.0
    exact_w = True
    for n, (s_h, z_h) in host["ptq_qparams"].items():
        s_d, z_d = qp[n]
        rel = float(((s_d - s_h).abs() / s_h.abs()).max())
        worst_s = max(worst_s, rel)
        worst_z = max(worst_z, float((z_d - z_h).abs().max()))
        if n.endswith("weight_quantizer"):
            exact_w = exact_w and bool(torch.equal(s_d, s_h))
    dy = float((y - host["ptq_logits"]).abs().max())
    print("PTQ vs host: scale max rel diff %.3e, zero_point max abs diff %.1f, weight scales bit-exact: %s, logits max abs diff %.3e (|logits| max %.3f)"
          % (worst_s, worst_z, exact_w, dy, float(host["ptq_logits"].abs().max())))
    assert sorted(qp) == sorted(host["ptq_qparams"]) and worst_s < 1e-4 and worst_z <= 1.0
    # ---- timing: the quantized forward, per-layer quantizer calls vs the model-wide weight launch ----
    x = batches(1, dev)[0]

    def timed(fn, n=50):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / n

    with torch.no_grad():
        ms_layer = timed(lambda: qm(x))
        from sparsebit_amd.group import WeightQuantGroup
        triples = [(m.weight_quantizer, m.weight, None) for _, m in qm.model.named_modules()
                   if getattr(m, "weight_quantizer", None) is not None and getattr(m.weight_quantizer, "is_enable", False)]
        grp = WeightQuantGroup(triples)
        handles = grp.attach(qm.model)
        y2 = qm(x)
        ms_group = timed(lambda: qm(x))
        for h in handles:
            h.remove()
        y1 = qm(x)
    print("quantized forward of the reference QuantModel on the device (batch 16): %.3f ms with one quantizer launch per layer, "
          "%.3f ms with WeightQuantGroup.attach (one launch for the %d weights); outputs identical: %s"
          % (ms_layer, ms_group, len(triples), bool(torch.equal(y1, y2))))
    # ---- round 5: the host out of the way -- launch plans (sparsebit_amd.plan) and one captured hipGraph of the forward
    #      (sparsebit_amd.graph) on the REAL QuantModel; the generic Python route per quantizer call is what round 4 shipped
    from sparsebit_amd import graph as sbq_graph
    from sparsebit_amd import plan as sbq_plan

    qm.eval()
    with torch.no_grad():
        sbq_plan.set_enabled(False)
        try:
            y_gen = qm(x).clone()
            ms_generic = timed(lambda: qm(x))
        finally:
            sbq_plan.set_enabled(True)
        y_plan = qm(x).clone()
        ms_plan = timed(lambda: qm(x))
    fwd = sbq_graph.capture(qm, x)
    y_graph = fwd(x).clone()
    ms_graph = timed(lambda: fwd(x))
    fwd_frozen = sbq_graph.capture(qm, x, freeze_weights=True)
    y_frozen = fwd_frozen(x).clone()
    ms_frozen = timed(lambda: fwd_frozen(x))
    for _, m in quantizers:
        m.disable_quant()
    fwd_float = sbq_graph.capture(qm, x)
    ms_float_graph = timed(lambda: fwd_float(x))
    with torch.no_grad():
        ms_float = timed(lambda: qm(x))
    for _, m in quantizers:
        if not m.fake_fused:
            m.enable_quant()
    print("quantized forward of the reference QuantModel, batch 16, host wall clock per forward: generic route %.3f ms, launch plans "
          "%.3f ms, captured graph %.3f ms (weights frozen into the capture: %.3f ms); float model (quantizers off) %.3f ms eager, "
          "%.3f ms as a graph; outputs identical to the generic route: plan %s, graph %s, frozen %s; graph speed-up %.2fx"
          % (ms_generic, ms_plan, ms_graph, ms_frozen, ms_float, ms_float_graph, bool(torch.equal(y_plan, y_gen)),
             bool(torch.equal(y_graph, y_gen)), bool(torch.equal(y_frozen, y_gen)), ms_generic / ms_graph))
    # ---- export: the reference's export_onnx loop (exporter stubbed), and the hand-written QDQ-ONNX file ----
    import torch.onnx

    seen = {}

    def fake_export(model, data, name, **kw):
        seen["flags"] = [m.export_onnx for _, m in quantizers]
        with torch.no_grad():
            seen["y"] = model(data)

    real = torch.onnx.export
    torch.onnx.export = fake_export
    try:
        qm.export_onnx(x.cpu(), os.path.join(tmp, "unused.onnx"))  # (the reference moves the model to the CPU, :245-246)
    finally:
        torch.onnx.export = real
    qm.model.to(dev)
    print("export_onnx loop: every quantizer in export mode during tracing: %s; export-branch output vs HIP output max abs diff %.3e"
          % (all(seen["flags"]), float((seen["y"].float().cpu() - y1.float().cpu()).abs().max())))
    from sparsebit_amd import export

    path = os.path.join(tmp, "resnet20_qdq.onnx")
    nbytes = export.save_qdq_onnx(qm.model, path)
    back = export.load_qdq_onnx(path)
    print("save_qdq_onnx: %d bytes, %d weight DequantizeLinear nodes, %d activation Q/DQ pairs, opset %d"
          % (nbytes, len(back["weights"]), len(back["activations"]), back["opset"]))
    # ---- QAT: LSQ 4w4a, one training step on the device ----
    lsq_yaml = open(os.path.join(ref, "examples", "quantization_aware_training", "cifar10", "basecase", "qconfig_lsq.yaml")).read()
    qm2 = build(ref, lsq_yaml, dev)
    for name in ("conv1", "fc"):
        getattr(qm2.model, name).input_quantizer.set_bit(bit=8)
        getattr(qm2.model, name).weight_quantizer.set_bit(bit=8)
    qm2.prepare_calibration()
    with torch.no_grad():
        for b in batches(4, dev):
            qm2(b)
        qm2.init_QAT()
    lsq = qparams(qm2)
    worst = 0.0
    for n, (s_h, _) in host["lsq_qparams"].items():
        worst = max(worst, float(((lsq[n][0].abs() - s_h.abs()).abs() / s_h.abs()).max()))
    print("LSQ init (lsq.py:32-51) vs host: %d quantizers, scale max rel diff %.3e" % (len(lsq), worst))
    assert worst < 1e-4
    qm2.train()
    opt = torch.optim.SGD(qm2.parameters(), 0.01, momentum=0.9)
    target = torch.randint(0, 10, (16,), generator=torch.Generator().manual_seed(2)).to(dev)
    losses = []
    for step in range(3):
        opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(qm2(x), target)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    scales = [m.scale for _, m in qm2.model.named_modules() if isinstance(m, rq.Quantizer) and isinstance(m.scale, torch.nn.Parameter)]
    with_grad = sum(1 for s in scales if s.grad is not None and bool(torch.isfinite(s.grad).all()) and float(s.grad.abs().sum()) > 0)
    print("LSQ QAT on the device: 3 SGD steps, losses %s, %d / %d learnable scales received a finite non-zero gradient"
          % (["%.4f" % v for v in losses], with_grad, len(scales)))
    assert all(v == v for v in losses) and with_grad == len(scales)
    print("OK: the reference's QuantModel ran on the MI355X through the plugin")


if __name__ == "__main__":
    main()
