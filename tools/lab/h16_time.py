"""Dev tool: k-th value / percentile of the headline tensor through the full-histogram engine (default) and round 3's
one-launch engine (knob 2 = 18), HIP events over rotating buffers; plus ReLU data and an extreme rank."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsebit_amd import lib as L  # noqa: E402
from sparsebit_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
w = (torch.randn(4096, 4096, generator=g) * torch.logspace(-2, 1, 4096).unsqueeze(1)).bfloat16()
xs = [torch.roll(w, i, 1).contiguous().to(dev) for i in range(12)]
relu = [torch.relu(x.float()).bfloat16() for x in xs]
n = w.numel()


def timed(fn, iters=200):
    for i in range(20):
        fn(i)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(iters):
            fn(i)
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / iters)
    return best


if os.environ.get("SBQ_LIB"):
    L.LIB_PATH = os.environ["SBQ_LIB"]  # a variant library (tools/lab/build_variant.py)
lib = L.load()
ws = L.fresh_workspace(lib.sbq_radix_select_workspace_bytes(1, 2), dev)
out = torch.empty(2, dtype=torch.float32, device=dev)
st = L.stream_ptr(dev)
import ctypes  # noqa: E402


def kth(data, k, use_abs):
    def f(i):
        lib.sbq_kth_value(L.ptr(data[i % 12]), L.BF16, n, use_abs, k, L.ptr(out), L.ptr(ws), ws.numel(), st)
    return f


def pct(data, alpha):
    ptrs = [(ctypes.c_void_p * 1)(d.data_ptr()) for d in data]
    outers = (ctypes.c_int64 * 1)(1)

    def f(i):
        lib.sbq_percentile_select(ptrs[i % 12], outers, 1, L.BF16, 1, n, alpha, L.ptr(out[0:1]), L.ptr(out[1:2]), L.ptr(ws), ws.numel(), st)
    return f


for knob in (0, 18):
    L.set_tuning(2, knob)
    print("knob2 =", knob, "(0: full-histogram engine, 18: win_one_kernel)")
    print("  kth |w| median          %.2f us" % timed(kth(xs, n // 2 + 1, 1)))
    print("  kth k=1                 %.2f us" % timed(kth(xs, 1, 0)))
    print("  kth k=n                 %.2f us" % timed(kth(xs, n, 0)))
    print("  percentile 1e-3         %.2f us" % timed(pct(xs, 1e-3)))
    print("  percentile 1e-5         %.2f us" % timed(pct(xs, 1e-5)))
    print("  percentile 0.2          %.2f us" % timed(pct(xs, 0.2)))
    print("  relu kth median         %.2f us" % timed(kth(relu, n // 2 + 1, 0)))
    print("  relu percentile 1e-3    %.2f us" % timed(pct(relu, 1e-3)))
    lib.sbq_kth_value(L.ptr(xs[0]), L.BF16, n, 1, n // 2 + 1, L.ptr(out), L.ptr(ws), ws.numel(), st)
    torch.cuda.synchronize()
    print("  value", float(out[0]), "ref", float(torch.sort(xs[0].float().abs().reshape(-1))[0][n // 2]))
L.set_tuning(2, 0)
