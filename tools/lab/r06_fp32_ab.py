"""lab: wall-clock of the three fp32 selections (model-wide thresholds, one 25.6 M kth value, four DeiT batches as
fp32) under a variant library:  SBQ_LIB=tools/lab/libsbq_variant.so python tools/lab/r06_fp32_ab.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

import bench_configs as B  # noqa: E402
from sparsebit_amd import lib as L  # noqa: E402

if os.environ.get("SBQ_LIB"):
    L.LIB_PATH = os.path.abspath(os.environ["SBQ_LIB"])
from sparsebit_amd import ops  # noqa: E402

print("library:", L.LIB_PATH, flush=True)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(50)
ws = [torch.randn(s, generator=g).to(dev) for s in B.resnet50_weight_shapes()]
ks = [min(int(w.numel() * 0.5), w.numel() - 1) + 1 for w in ws]
big = torch.randn(25_600_000, generator=g).to(dev)
acts = [torch.randn(64 * 197 * 384, generator=g).to(dev) for _ in range(4)]


def timed(fn, iters=300):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


ref_g = torch.stack([ops.kth_value(w, k, True) for w, k in zip(ws, ks)])
assert torch.equal(ops.group_kth_value(ws, ks, True), ref_g)
acts16 = [a.bfloat16() for a in acts]
w16 = (torch.randn(4096, 4096, generator=g) * torch.logspace(-2, 1, 4096).unsqueeze(1)).to(dev).bfloat16()
big16 = torch.randn(25_600_000, generator=g).to(dev).bfloat16()
huge16 = torch.randn(40_000_000, generator=g).to(dev).bfloat16()
for rep in range(3):
    print("model-wide thresholds %.1f us | kth 25.6M fp32 %.1f us | kth 32 K fp32 (one workgroup) %.1f us | percentile 4 x 4.8M fp32 %.1f us | kth 25.6M "
          "bf16 %.1f us | kth 40M bf16 %.1f us | percentile 4 x 4.8M bf16 %.1f us | percentile 40M bf16 %.1f us | 4096^2 bf16 (full-histogram engine): kth %.1f us, percentile %.1f us" % (
        timed(lambda: ops.group_kth_value(ws, ks, True)),
        timed(lambda: ops.kth_value(big, 12_800_000, True)),
        timed(lambda: ops.kth_value(ws[10], ks[10], True)),
        timed(lambda: ops.percentile_select(acts, 1e-3, 0, False), 100),
        timed(lambda: ops.kth_value(big16, 12_800_000, True)),
        timed(lambda: ops.kth_value(huge16, 20_000_000, True)),
        timed(lambda: ops.percentile_select(acts16, 1e-3, 0, False)),
        timed(lambda: ops.percentile_select([huge16], 1e-3, 0, False)),
        timed(lambda: ops.kth_value(w16, 8_388_609, True)),
        timed(lambda: ops.percentile_select([w16], 1e-3, 0, False))), flush=True)
