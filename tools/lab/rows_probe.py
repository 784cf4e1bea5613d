"""Dev tool: per-channel percentile of a 4096 x 4096 weight (sbq_percentile_rows), packed top-R lists vs the extraction."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from sparsebit_amd import lib as L, ops
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream(dev)
def timed(fn, iters=200, warm=20, rounds=3):
    best = 1e9
    for _ in range(rounds):
        for i in range(warm): fn(i)
        torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for i in range(iters): fn(i)
        b.record(stream); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / iters)
    return best
lib = L.load(); st = L.stream_ptr(dev)
g = torch.Generator().manual_seed(0)
for dt, did in ((torch.bfloat16, L.BF16), (torch.float32, L.F32)):
    xs = [(torch.randn(4096, 4096, generator=g) * torch.logspace(-2, 1, 4096).unsqueeze(1)).to(dt).to(dev) for _ in range(12)]
    mn = torch.empty(4096, dtype=torch.float32, device=dev); mx = torch.empty_like(mn)
    for alpha in (1e-3, 5e-4, 1.9e-3):
        row = []
        for knob in (0, 17):
            L.set_tuning(2, knob)
            row.append(timed(lambda i: lib.sbq_percentile_rows(L.ptr(xs[i % 12]), did, 4096, 4096, alpha, L.ptr(mn), L.ptr(mx), st)))
        L.set_tuning(2, 0)
        print("%s alpha %g: packed lists %.2f us, extraction %.2f us" % (dt, alpha, row[0], row[1]), flush=True)
