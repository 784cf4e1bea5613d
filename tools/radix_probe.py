"""Dev tool: time of each radix-select pass (histogram kernel) on a 4096x4096 bf16 tensor, per tensor."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sparsebit_amd import ops, lib as L
dev = torch.device("cuda:0")
x = torch.randn(4096, 4096, device=dev).bfloat16()
be = ops.HipSelectBackend()
def timed(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters
for n_sel, use_abs in ((2, False), (1, True)):
    ranks = [[1000, 16000000][:n_sel]]
    state = be.new_state(ranks, dev)
    for p in range(3):
        hist = be.new_hist(1, n_sel, dev)
        t = timed(lambda: be.histogram(x, state, hist, p, n_sel, use_abs, 0, False))
        hist.zero_(); be.histogram(x, state, hist, p, n_sel, use_abs, 0, False)
        nz = int((hist[0, 0] > 0).sum())
        ta = timed(lambda: be.advance(hist, state.clone(), p, n_sel, 1))
        be.advance(hist, state, p, n_sel, 1)
        print("n_sel=%d abs=%d pass %d: histogram %.1f us (%d non-empty bins), advance %.1f us" % (n_sel, use_abs, p, t, nz, ta), flush=True)
print("kth_value e2e: %.1f us" % timed(lambda: ops.kth_value(x, 8000000, True)))
print("percentile_select e2e: %.1f us" % timed(lambda: ops.percentile_select([x], 1e-3, 0, False)))
