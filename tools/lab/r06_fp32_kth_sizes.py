"""lab: per-call time of the one-launch fp32 k-th value over many distinct tensors of one size (which inputs are slow?)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from sparsebit_amd import lib as L, ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
n = int(os.environ.get("N", 1 << 20))
# (the same generator stream as tools/r06_fp32_kth_probe.py up to its third size)
for m in (4096 * 4096, 2359296):
    for _ in range(max(2, (320 << 20) // (4 * m))):
        torch.randn(m, generator=g)
xs = [(torch.randn(n, generator=g) * 0.05).to(dev) for _ in range(80)]
k = n // 2
slow = []
for rep in range(2):
    for i, x in enumerate(xs):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); v = ops.kth_value(x, k, True); b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3
        if us > 100:
            want = float(torch.sort(x.abs())[0][k - 1])
            slow.append((rep, i, round(us, 1), float(v) == want))
print("n = %d: %d slow calls of %d: %r" % (n, len(slow), 2 * len(xs), slow[:12]))
# back to back, no synchronisation in between
for i in range(10):
    ops.kth_value(xs[i], k, True)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for i in range(100):
    ops.kth_value(xs[i % 80], k, True)
b.record(); torch.cuda.synchronize()
print("back to back over the 80 tensors: %.1f us per call" % (a.elapsed_time(b) * 10.0))
a.record()
for i in range(100):
    ops.kth_value(xs[0], k, True)
b.record(); torch.cuda.synchronize()
print("back to back on ONE tensor: %.1f us per call" % (a.elapsed_time(b) * 10.0))
