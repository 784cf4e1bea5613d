"""A ResNet-20 (CIFAR) written in the reference's QuantOpr convention -- every convolution / linear / residual add
quantizes its float inputs with `input_quantizer` and its weight with `weight_quantizer` inline, as
sparsebit/quantization/modules/conv.py:30-42, linear.py:21-34 and math.py (QAdd) do after BN fusion -- so that the
end-to-end cost of a quantized forward can be measured on a box that has no reference checkout
(bench_configs.e2e_resnet20; the same three numbers on the REAL QuantModel(resnet20): tools/run_reference_on_device.py).

    python examples/resnet20_quantopr.py        # eager vs launch plans vs captured graph, batch 16 (needs an MI355X)

W per-channel-symmetric int8, A per-tensor-affine uint8, min-max observers (the PTQ config of
examples/post_training_quantization/imagenet1k/basecase at CIFAR size): 22 operators with weights + 9 adds = 62
quantizer calls per forward.
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from sparsebit_amd.common import Backend  # noqa: E402
from sparsebit_amd.config import quantizer_config  # noqa: E402
from sparsebit_amd.quantizers import build_quantizer  # noqa: E402


def _wq():
    q = build_quantizer(quantizer_config("per-channel-symmetric", 8, observer="MINMAX", target="weight"))
    q.set_backend(Backend.VIRTUAL)
    return q


def _aq():
    q = build_quantizer(quantizer_config("per-tensor-affine", 8, observer="MINMAX", target="feature", layout="NCHW"))
    q.set_backend(Backend.VIRTUAL)
    return q


class QConv2d(torch.nn.Module):
    def __init__(self, cin, cout, k, stride=1):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.randn(cout, cin, k, k) * (2.0 / (cin * k * k)) ** 0.5)
        self.bias = torch.nn.Parameter(torch.zeros(cout))  # (the folded BN)
        self.stride, self.pad = stride, k // 2
        self.input_quantizer, self.weight_quantizer = _aq(), _wq()

    def forward(self, x):
        return F.conv2d(self.input_quantizer(x), self.weight_quantizer(self.weight), self.bias, self.stride, self.pad)


class QLinear(torch.nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.randn(cout, cin) * (1.0 / cin) ** 0.5)
        self.bias = torch.nn.Parameter(torch.zeros(cout))
        self.input_quantizer, self.weight_quantizer = _aq(), _wq()

    def forward(self, x):
        return F.linear(self.input_quantizer(x), self.weight_quantizer(self.weight), self.bias)


class QAdd(torch.nn.Module):
    """both addends through ONE input quantizer, like the reference's QAdd"""

    def __init__(self):
        super().__init__()
        self.input_quantizer = _aq()

    def forward(self, a, b):
        return self.input_quantizer(a) + self.input_quantizer(b)


class Block(torch.nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.c1, self.c2 = QConv2d(cin, cout, 3, stride), QConv2d(cout, cout, 3)
        self.short = QConv2d(cin, cout, 1, stride) if (stride != 1 or cin != cout) else None
        self.add = QAdd()

    def forward(self, x):
        y = self.c2(torch.relu(self.c1(x)))
        return torch.relu(self.add(y, x if self.short is None else self.short(x)))


class ResNet20(torch.nn.Module):
    def __init__(self, classes=10):
        super().__init__()
        self.stem = QConv2d(3, 16, 3)
        blocks, cin = [], 16
        for cout, stride in ((16, 1), (32, 2), (64, 2)):
            for b in range(3):
                blocks.append(Block(cin, cout, stride if b == 0 else 1))
                cin = cout
        self.blocks = torch.nn.Sequential(*blocks)
        self.fc = QLinear(64, classes)

    def forward(self, x):
        x = self.blocks(torch.relu(self.stem(x)))
        return self.fc(x.mean(dim=(2, 3)))


def quantizers(model):
    out = []
    for m in model.modules():
        for name in ("input_quantizer", "weight_quantizer"):
            q = getattr(m, name, None)
            if q is not None:
                out.append(q)
    return out


def build(device, seed=0, batches=2, batch=16):
    """-> (calibrated eval-mode model with every quantizer enabled, example input)"""
    from sparsebit_amd.calibration import DeviceCalibrator

    torch.manual_seed(seed)
    model = ResNet20().to(device).eval()
    g = torch.Generator().manual_seed(seed + 1)
    data = [torch.randn(batch, 3, 32, 32, generator=g).to(device) for _ in range(batches)]
    DeviceCalibrator(model).calibrate(data)
    for q in quantizers(model):
        q.enable_quant()
    return model, data[0]


def time_forward(fn, iters=200, warm=20):
    """host wall clock per call with the device drained at both ends: what a caller of model(x) waits for"""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6 / iters


def measure(device, iters=200):
    """eager (generic Python route per quantizer call) vs launch plans vs one captured graph; outputs compared bit for bit"""
    from sparsebit_amd import graph as sbq_graph
    from sparsebit_amd import plan as sbq_plan

    model, x = build(device)
    n_q = len(quantizers(model))
    with torch.no_grad():
        sbq_plan.set_enabled(False)
        try:
            y_eager = model(x).clone()
            eager_us = time_forward(lambda: model(x), iters)
        finally:
            sbq_plan.set_enabled(True)
        y_plan = model(x).clone()
        plan_us = time_forward(lambda: model(x), iters)
    fwd = sbq_graph.capture(model, x)
    y_graph = fwd(x).clone()
    graph_us = time_forward(lambda: fwd(x), iters)
    fwd_frozen = sbq_graph.capture(model, x, freeze_weights=True)
    y_frozen = fwd_frozen(x).clone()
    frozen_us = time_forward(lambda: fwd_frozen(x), iters)
    # the float model (quantizers off): what the layers themselves cost -- through the same Python, and as a graph of its
    # own (the floor no quantizer path can go below: torch's / MIOpen's kernels and their boundaries)
    for q in quantizers(model):
        q.disable_quant()
    with torch.no_grad():
        float_us = time_forward(lambda: model(x), iters)
    fwd_float = sbq_graph.capture(model, x)
    float_graph_us = time_forward(lambda: fwd_float(x), iters)
    for q in quantizers(model):
        q.enable_quant()
    return {
        "eager_us": round(eager_us, 1), "plan_us": round(plan_us, 1), "graph_us": round(graph_us, 1),
        "graph_frozen_weights_us": round(frozen_us, 1),
        "float_model_eager_us": round(float_us, 1), "float_model_graph_us": round(float_graph_us, 1),
        "quantizers_cost_in_graph_us": round(graph_us - float_graph_us, 1),
        "quantizers_cost_in_graph_frozen_weights_us": round(frozen_us - float_graph_us, 1),
        "frozen_equals_eager": bool(torch.equal(y_frozen, y_eager)),
        "quantizer_calls_per_forward": n_q + 9,  # (a QAdd's quantizer runs twice)
        "plan_equals_eager": bool(torch.equal(y_plan, y_eager)), "graph_equals_eager": bool(torch.equal(y_graph, y_eager)),
        "graph_speedup_vs_eager": round(eager_us / graph_us, 2),
    }


if __name__ == "__main__":
    import json

    print(json.dumps(measure(torch.device("cuda:0"))))
