"""Golden vectors for the calibration DRIVER, from the REAL reference (authoring container only).

Builds a reference QuantModel (fx trace, BN fusion, quantizer placement, DISABLE_UNNECESSARY_QUANT) around
a small CNN and runs the reference's own CalibrationRunner (sparsebit/quantization/tools/calibration.py:
11-160) on the CPU in BOTH modes -- asym=False and asym=True with w_quant=a_quant=True -- for two observer
configurations.  Stores the post-fusion operator weights, the calibration batches, and every quantizer's
scale / zero_point in tests/golden/calib_golden.npz.  tests/test_gpu_plugin.py::test_device_calibrator_equals_reference_calibration_runner rebuilds the same operator
chain on the GPU box (where the reference does not exist) and checks sparsebit_amd.calibration.
DeviceCalibrator against these numbers.

Re-run:  HIP_VISIBLE_DEVICES="" python tests/golden/gen_calib_golden.py
"""
import os
import sys
import tempfile

os.environ.setdefault("HIP_VISIBLE_DEVICES", "")
os.environ.setdefault("CUDA_VISIBLE_DEVICES", "")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

OUT = os.path.join(HERE, "calib_golden.npz")

CONFIGS = {
    # name: (W scheme, W observer, A scheme, A observer)
    "minmax": ("per-channel-symmetric", "MINMAX", "per-tensor-affine", "MINMAX"),
    "pct_mse": ("per-channel-symmetric", "PERCENTILE", "per-tensor-symmetric", "MSE"),
}

CFG = """
BACKEND: virtual
SCHEDULE:
  FUSE_BN: True
W:
  QSCHEME: {ws}
  QUANTIZER:
    TYPE: uniform
    BIT: 8
  OBSERVER:
    TYPE: {wo}
A:
  QSCHEME: {as_}
  QUANTIZER:
    TYPE: uniform
    BIT: 8
  OBSERVER:
    TYPE: {ao}
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


def main():
    assert not torch.cuda.is_available(), "generate goldens with GPUs hidden (reference CPU path)"
    gen_golden.install_stubs()
    sys.path.insert(0, gen_golden.REF)
    from sparsebit.quantization import QuantModel, parse_qconfig
    from sparsebit.quantization.quantizers import Quantizer

    out = {}
    g = torch.Generator().manual_seed(77)
    batches = [torch.randn(4, 3, 8, 8, generator=g) * (1.0 + 0.5 * i) for i in range(3)]
    out["batches"] = np.stack([b.numpy() for b in batches])
    for name, (ws, wo, as_, ao) in CONFIGS.items():
        for asym in (False, True):
            torch.manual_seed(5)
            net = Net()
            g = torch.Generator().manual_seed(78)
            with torch.no_grad():  # non-trivial BN statistics so that fusion changes the weights
                net.bn.running_mean.copy_(torch.randn(8, generator=g) * 0.1)
                net.bn.running_var.copy_(torch.rand(8, generator=g) + 0.5)
                net.bn.weight.copy_(torch.rand(8, generator=g) + 0.5)
                net.bn.bias.copy_(torch.randn(8, generator=g) * 0.1)
            net.eval()
            with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
                f.write(CFG.format(ws=ws, wo=wo, as_=as_, ao=ao))
            qm = QuantModel(net, parse_qconfig(f.name))
            os.unlink(f.name)
            qm.prepare_calibration()
            with torch.no_grad():
                for b in batches:
                    qm(b)
            qm.calc_qparams(asym=asym, w_quant=asym, a_quant=asym)
            tag = "{}/asym{}".format(name, int(asym))
            names = []
            for n, m in qm.model.named_modules():
                if isinstance(m, Quantizer):
                    names.append(n)
                    out["{}/{}/scale".format(tag, n)] = m.scale.detach().reshape(-1).numpy().copy()
                    out["{}/{}/zero_point".format(tag, n)] = m.zero_point.detach().reshape(-1).numpy().copy()
                    out["{}/{}/fake_fused".format(tag, n)] = np.array(int(m.fake_fused))
            out[tag + "/names"] = np.array(names)
            if not asym:
                mods = dict(qm.model.named_modules())
                for n in ("c1_bn", "c2", "fc"):
                    out["{}/{}/weight".format(name, n)] = mods[n].weight.detach().numpy().copy()
                    out["{}/{}/bias".format(name, n)] = mods[n].bias.detach().numpy().copy()
                # quantized end-to-end output of the calibrated model on batch 0 (CPU reference path)
                qm.set_quant(w_quant=True, a_quant=True)
                with torch.no_grad():
                    out[name + "/y_quant"] = qm(batches[0]).numpy().copy()
                    qm.set_quant(False, False)
                    out[name + "/y_float"] = qm(batches[0]).numpy().copy()
        # the reference's two modes agree on every qparam (run_feature_calibration always reads the float storage)
        for n in out[name + "/asym0/names"]:
            for k in ("scale", "zero_point"):
                a, b = out["{}/asym0/{}/{}".format(name, n, k)], out["{}/asym1/{}/{}".format(name, n, k)]
                assert np.array_equal(a, b), (name, n, k)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, len(out), "arrays")


if __name__ == "__main__":
    main()
