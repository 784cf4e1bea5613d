"""Round 6: the model-wide MSE calibration (53 fp32 ResNet-50 weights, per channel, 8 bit symmetric) -- a lane per
(row, candidate) (calib_mse_lanes_kernel) against round 3's wave-per-row form (knob 2 = 37)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench_configs as B  # noqa: E402
from sparsebit_amd import lib as L  # noqa: E402
from sparsebit_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(50)
ws = [torch.randn(s, generator=g).to(dev) for s in B.resnet50_weight_shapes()]
n = sum(w.numel() for w in ws)
res = {}
for knob, name in ((0, "lane per (row, candidate)"), (37, "round 3: wave per row")):
    L.set_tuning(2, knob)
    grp = ops.GroupCalibration([(w, -128, 127, True, True) for w in ws])
    L.set_tuning(2, 0)
    for _ in range(3):
        grp.launch_mse()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        grp.launch_mse()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / 20
    res[knob] = torch.cat([v.clone() for v in grp.views["index"]])
    print("%-28s %7.1f us  (min-max: 2 launches; search: 1 + the pick inside for the lane form, 2 for round 3)  %.1f TFLOP/s at 7 flop per evaluation" % (name, us, n * 80 * 7 / us / 1e6), flush=True)
print("rows: %d, argmin index differs in %d" % (res[0].numel(), int((res[0] != res[37]).sum())))
