"""Dev tool: end-to-end time of the order-statistics entry points (events over back-to-back calls, C ABI through
ops.py; the workspace is cached, so a call is only its launches).  knob 2 = 7: fixed-digit radix engine."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sparsebit_amd import lib as L, ops
dev = torch.device("cuda:0")
QUICK = "--quick" in sys.argv  # profiling runs: one short round per entry point
def timed(fn, iters=50, warm=10):
    best = 1e9
    if QUICK: iters, warm = 10, 3
    for _ in range(1 if QUICK else 3):
        for _ in range(warm): fn()
        torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters): fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / iters)
    return best
g = torch.Generator().manual_seed(0)
w = (torch.randn(4096, 4096, generator=g) * torch.logspace(-2, 1, 4096).unsqueeze(1))
for dt in (torch.bfloat16, torch.float32):
    x = w.to(dt).to(dev)
    n = x.numel()
    for knob in (7, 0):
        L.set_tuning(2, knob)
        t1 = timed(lambda: ops.kth_value(x, n // 2 + 1, True))
        t2 = timed(lambda: ops.percentile_select([x], 1e-3, 0, False))
        print("%-9s engine=%s  kth_value(|x|, 50%%) %7.1f us   percentile per tensor %7.1f us" % (str(dt)[6:], "fixed-digit" if knob == 7 else "windowed", t1, t2), flush=True)
    L.set_tuning(2, 0)
    t3 = timed(lambda: ops.percentile_rows(x, 1e-3))
    print("%-9s percentile_rows (per channel, 4096 rows) %7.1f us" % (str(dt)[6:], t3), flush=True)
acts = [torch.randn(64, 197, 384, generator=g).bfloat16().to(dev) for _ in range(4)]
for knob in (7, 0):
    L.set_tuning(2, knob)
    t = timed(lambda: ops.percentile_select(acts, 1e-3, 0, False))
    print("DeiT activations 4 x 64x197x384 bf16 per tensor, engine=%s: %7.1f us" % ("fixed-digit" if knob == 7 else "windowed", t), flush=True)
L.set_tuning(2, 0)
# a window the sample misses: every sampled pack holds an outlier (period == sampling stride), so the first sweep
# finds the rank outside its window and the remaining rounds run inside the one fallback launch
n = 4096 * 4096
per = torch.randn(n, generator=g)
stride = n // 2048
idx = (torch.arange(2048) * stride).unsqueeze(1) + torch.arange(8).unsqueeze(0)
per[idx.reshape(-1)] = 1000.0
xb = per.bfloat16().to(dev)
t = timed(lambda: ops.kth_value(xb, n // 2 + 1, True))
print("bfloat16  kth_value with a missed first window (sampled packs are outliers): %7.1f us" % t, flush=True)
