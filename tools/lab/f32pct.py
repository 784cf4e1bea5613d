import sys, torch
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
w = (torch.randn(4096, 4096, generator=g) * torch.logspace(-2, 1, 4096).unsqueeze(1)).to(dev)
acts = [torch.randn(64, 197, 384, generator=g).to(dev) for _ in range(4)]
for knob in (0, 3):
    L.set_tuning(2, knob)
    r1 = ops.percentile_select([w], 1e-3, 0, False)
    t1 = timed(lambda: ops.percentile_select([w], 1e-3, 0, False))
    r2 = ops.percentile_select(acts, 1e-3, 0, False)
    t2 = timed(lambda: ops.percentile_select(acts, 1e-3, 0, False))
    print("knob", knob, "fp32 weight percentile %.1f us" % t1, [float(v) for v in r1], " DeiT fp32 %.1f us" % t2, [float(v) for v in r2], flush=True)
L.set_tuning(2, 0)
