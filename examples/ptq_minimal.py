"""Post-training quantization of a small MLP, entirely on the GPU (needs an MI355X).

Shows the pieces of the hot path working together through the reference-style API:
quantizer / observer registries, the device-resident calibration driver, the fake-quant
forward, and the real int8 tensors the same kernel can emit.

    python examples/ptq_minimal.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sparsebit_amd.calibration import DeviceCalibrator  # noqa: E402
from sparsebit_amd.common import Backend  # noqa: E402
from sparsebit_amd.config import quantizer_config  # noqa: E402
from sparsebit_amd.quantizers import build_quantizer  # noqa: E402


class QLinear(torch.nn.Module):
    """A quantized operator in the reference's QuantOpr convention (modules/linear.py:30-34)."""

    def __init__(self, lin, observer):
        super().__init__()
        self.weight, self.bias = lin.weight, lin.bias
        self.weight_quantizer = build_quantizer(quantizer_config("per-channel-symmetric", 8))
        self.input_quantizer = build_quantizer(
            quantizer_config("per-tensor-affine", 8, observer=observer, target="feature", layout="NLC"))
        for q in (self.weight_quantizer, self.input_quantizer):
            q.set_backend(Backend.VIRTUAL)

    def forward(self, x):
        return torch.nn.functional.linear(self.input_quantizer(x), self.weight_quantizer(self.weight), self.bias)


def main():
    torch.manual_seed(0)
    dev = "cuda"
    f1, f2 = torch.nn.Linear(384, 1536), torch.nn.Linear(1536, 384)
    float_model = torch.nn.Sequential(f1, torch.nn.GELU(), f2).to(dev)
    for observer in ("MINMAX", "PERCENTILE", "MSE"):
        qmodel = torch.nn.Sequential(QLinear(f1, observer), torch.nn.GELU(), QLinear(f2, observer)).to(dev)
        calib = [torch.randn(64, 197, 384, device=dev) for _ in range(4)]
        qparams = DeviceCalibrator(qmodel).calibrate(calib)
        for m in qmodel:
            if isinstance(m, QLinear):
                m.input_quantizer.enable_quant()
                m.weight_quantizer.enable_quant()
        x = torch.randn(64, 197, 384, device=dev)
        with torch.no_grad():
            err = (qmodel(x) - float_model(x)).pow(2).mean().sqrt() / float_model(x).pow(2).mean().sqrt()
        print("%-10s observer: %d quantizers calibrated, relative output error of the 8w8a model %.4f"
              % (observer, len(qparams), err.item()))
    wq = qmodel[0].weight_quantizer
    dq, q_int = wq.quantize_to_int(qmodel[0].weight.detach())
    print("weight as real int8:", tuple(q_int.shape), q_int.dtype, "range", int(q_int.min()), int(q_int.max()),
          "| dq == q * scale:", bool(torch.equal(dq, q_int.float() * wq.scale)))


if __name__ == "__main__":
    main()
