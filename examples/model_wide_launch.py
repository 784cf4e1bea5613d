"""All weight quantizers of a small CNN in one launch per step, attached to unmodified operators, then the
weights exported as real integer tensors (needs an MI355X).

    python examples/model_wide_launch.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sparsebit_amd import export  # noqa: E402
from sparsebit_amd.common import Backend  # noqa: E402
from sparsebit_amd.config import quantizer_config  # noqa: E402
from sparsebit_amd.group import WeightQuantGroup  # noqa: E402
from sparsebit_amd.quantizers import build_quantizer  # noqa: E402


class QConv(torch.nn.Module):
    """written like the reference's QConv2d: the weight quantizer is called inline (modules/conv.py:30-36)"""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.randn(cout, cin, k, k) * (2.0 / (cin * k * k)) ** 0.5)
        self.weight_quantizer = build_quantizer(quantizer_config("per-channel-symmetric", 4, quantizer="lsq"))
        self.weight_quantizer.set_backend(Backend.VIRTUAL)
        self.pad = k // 2

    def forward(self, x):
        return torch.nn.functional.conv2d(x, self.weight_quantizer(self.weight), padding=self.pad)


def main():
    torch.manual_seed(0)
    layers = []
    cin = 16
    for cout in (32, 32, 64, 64, 128, 128, 128, 128, 256, 256, 256, 256):
        layers += [QConv(cin, cout, 3), torch.nn.ReLU()]
        cin = cout
    model = torch.nn.Sequential(QConv(8, 16, 3), torch.nn.ReLU(), *layers).cuda()
    oprs = [m for m in model if isinstance(m, QConv)]
    for m in oprs:  # LSQ initialises its step sizes from the weights
        m.weight_quantizer.update_observer(m.weight.detach())
        m.weight_quantizer.calc_qparams()
        m.weight_quantizer.enable_quant()
    params = [p for m in oprs for p in (m.weight, m.weight_quantizer.scale)]
    opt = torch.optim.SGD(params, lr=1e-3)
    x = torch.randn(8, 8, 32, 32, device="cuda")

    def steps(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            opt.zero_grad(set_to_none=True)
            loss = model(x).square().mean()
            loss.backward()
            opt.step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, loss.item()

    steps(3)
    t_layer, l0 = steps(20)
    group = WeightQuantGroup([(m.weight_quantizer, m.weight, None) for m in oprs])
    handles = group.attach(model)  # zero edits to QConv: its inline quantizer call now returns the grouped result
    steps(3)
    t_group, l1 = steps(20)
    print("%d weight quantizers, QAT step: %.2f ms one launch per layer, %.2f ms with the model-wide launch"
          % (len(oprs), t_layer, t_group))
    for h in handles:
        h.remove()
    # export: int4 levels + scale + zero point of one layer, straight from the fake-quant launch
    m = oprs[3]
    dq, rec = export.quantize_linear(m.weight_quantizer, m.weight.detach(), pack_int4=True)
    print("layer 3 as packed int4: %d bytes for %d weights, bits=%d, axis=%s; DequantizeLinear == fake-quant: %s"
          % (rec.q.numel(), m.weight.numel(), rec.bits, rec.axis, bool(torch.equal(rec.dequantize(), dq))))


if __name__ == "__main__":
    main()
