"""Dev tool: selection timings over ranks / alphas / data shapes (robustness of the one-launch engine's plan)."""
import os, sys, ctypes
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from sparsebit_amd import lib as L, ops
lib = L.load()
dev = torch.device("cuda:0")
st = L.stream_ptr(dev)
stream = torch.cuda.current_stream(dev)
def timed(fn, iters=50, warm=5, rounds=2):
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
n = 4096 * 4096
g = torch.Generator().manual_seed(0)
data = {
    "gauss": torch.randn(n, generator=g),
    "relu": torch.relu(torch.randn(n, generator=g)),
    "logscale": (torch.randn(4096, 4096, generator=g) * torch.logspace(-2, 1, 4096).unsqueeze(1)).reshape(-1),
    "smooth": torch.cumsum(torch.randn(n, generator=g), 0) * 1e-2,
}
sw = torch.zeros(lib.sbq_radix_select_workspace_bytes(1, 2), dtype=torch.uint8, device=dev)
out = torch.empty(2, dtype=torch.float32, device=dev)
for dt, did in ((torch.bfloat16, L.BF16), (torch.float16, L.F16), (torch.float32, L.F32)):
    for name, x in data.items():
        xs = [x.to(dt).to(dev)] + [torch.roll(x, 1000 * i).to(dt).to(dev) for i in range(1, 4)]
        row = []
        for k in (1, n // 1000, n // 10, n // 2, n - n // 1000, n):
            for ua in (0, 1):
                row.append("%.0f" % timed(lambda i: lib.sbq_kth_value(L.ptr(xs[i % 4]), did, n, ua, k, L.ptr(out), L.ptr(sw), sw.numel(), st)))
        prow = []
        for alpha in (1e-1, 1e-2, 1e-3, 1e-4, 1e-5, 1e-6):
            def pct(i):
                p = (ctypes.c_void_p * 1)(xs[i % 4].data_ptr()); o = (ctypes.c_int64 * 1)(1)
                lib.sbq_percentile_select(p, o, 1, did, 1, n, alpha, L.ptr(out[0:1]), L.ptr(out[1:2]), L.ptr(sw), sw.numel(), st)
            prow.append("%.0f" % timed(pct))
        print("%-9s %-9s kth us (k=1,n/1000,n/10,n/2,n-n/1000,n; x / |x|): %s | pct us (alpha 1e-1..1e-6): %s" % (
            str(dt).replace("torch.", ""), name, " ".join(row), " ".join(prow)), flush=True)
        del xs
