"""Round 6: the model-wide L1 thresholds (53 fp32 ResNet-50 weights, ratio 0.5; bench_configs.model_wide_calibration) --
the grouped selection with candidate segments (one launch) against round 5's two launches (knob 2 = 34).
  python tools/r06_group_probe.py            # wall-clock per call + equality
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench_configs as B  # noqa: E402
from sparsebit_amd import lib as L  # noqa: E402
from sparsebit_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(50)
ws = [torch.randn(s, generator=g).to(dev) for s in B.resnet50_weight_shapes()]
ks = [min(int(w.numel() * 0.5), w.numel() - 1) + 1 for w in ws]
nbytes = sum(w.numel() for w in ws) * 4


def timed(iters):
    for _ in range(5):
        ops.group_kth_value(ws, ks, True)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        ops.group_kth_value(ws, ks, True)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


ref = torch.stack([ops.kth_value(w, k, True) for w, k in zip(ws, ks)])
for knob, name in ((0, "candidate store in LDS, 1 launch"), (34, "round 5: 2 launches")):
    L.set_tuning(2, knob)
    got = ops.group_kth_value(ws, ks, True)
    us = timed(200)
    L.set_tuning(2, 0)
    print("%-32s %7.1f us  %6.1f GB/s algorithmic  == per-tensor: %s" % (name, us, nbytes / us / 1e3, bool(torch.equal(got, ref))), flush=True)
