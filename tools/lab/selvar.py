import os, sys, torch
sys.path.insert(0, "/root/repo")
from sparsebit_amd import lib as L, ops
dev = torch.device("cuda:0")
def timed(fn, iters=50, warm=10):
    best = 1e9
    for _ in range(3):
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
    x = w.to(dt).to(dev); n = x.numel()
    for knob in (0, 8):
        L.set_tuning(2, knob)
        t1 = timed(lambda: ops.kth_value(x, n // 2 + 1, True))
        t2 = timed(lambda: ops.percentile_select([x], 1e-3, 0, False))
        print(dt, "knob", knob, "kth %.1f pct %.1f" % (t1, t2), flush=True)
L.set_tuning(2, 0)
