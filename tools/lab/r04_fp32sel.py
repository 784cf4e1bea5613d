"""Dev tool: fp32 selections with two launches (knob 2 = 20: the second one resident if a third sweep is needed) against
the three launches of round 3 -- single tensor and the model-wide L1 thresholds."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench_configs as BC  # noqa: E402
from sparsebit_amd import lib as L  # noqa: E402
from sparsebit_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / iters)
    return best


g = torch.Generator().manual_seed(50)
ws = [torch.randn(shp, generator=g).to(dev) for shp in BC.resnet50_weight_shapes()]
ks = [min(int(w.numel() * 0.5), w.numel() - 1) + 1 for w in ws]
big = (torch.randn(4096, 4096, generator=g) * torch.logspace(-2, 1, 4096).unsqueeze(1)).to(dev)
acts = torch.relu(torch.randn(64, 197, 1536, generator=g)).to(dev)
ref = None
for knob in (0, 20):
    L.set_tuning(2, knob)
    got = ops.group_kth_value(ws, ks, True)
    v1 = ops.kth_value(big, big.numel() // 2 + 1, True)
    p1 = ops.percentile_select([big], 1e-3, 0, False)
    p2 = ops.percentile_select([acts], 1e-3, 0, False)
    torch.cuda.synchronize()
    cur = (got.clone(), float(v1), float(p1[0]), float(p1[1]), float(p2[0]), float(p2[1]))
    if ref is None:
        ref = cur
    same = torch.equal(cur[0], ref[0]) and cur[1:] == ref[1:]
    print("knob2=%d: 53 L1 thresholds %.1f us, kth fp32 16.7M %.1f us, percentile fp32 16.7M %.1f us, percentile relu 19.4M %.1f us, same results: %s" % (
        knob, timed(lambda: ops.group_kth_value(ws, ks, True)), timed(lambda: ops.kth_value(big, big.numel() // 2 + 1, True)),
        timed(lambda: ops.percentile_select([big], 1e-3, 0, False)), timed(lambda: ops.percentile_select([acts], 1e-3, 0, False)), same))
L.set_tuning(2, 0)
