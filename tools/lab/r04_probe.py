"""Dev tool (round 4): (a) the min-max statistics kernels on fp32 tensors of several shapes -- why is the per-tensor
activation leg at 3.3 TB/s when the per-channel weight runs at 5.3?  (b) the per-tensor MSE routes.  Run under
rocprofv3 --kernel-trace --stats for the per-kernel breakdown."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsebit_amd import lib as L  # noqa: E402
from sparsebit_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
lib = L.load()
st = L.stream_ptr(dev)


def timed(fn, iters=50):
    for i in range(5):
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


g = torch.Generator().manual_seed(0)
for name, shape, relu in (("randn 4096x4096", (4096, 4096), False), ("randn 64x64x56x56", (64, 64, 56, 56), False),
                          ("relu 64x64x56x56", (64, 64, 56, 56), True), ("randn 25.7M flat", (25690112,), False)):
    n = 1
    for d in shape:
        n *= d
    copies = max(2, int(6e8 // (n * 4)) + 1)
    base = torch.randn(n, generator=g)
    if relu:
        base = torch.relu(base)
    xs = [torch.roll(base, i).contiguous().to(dev) for i in range(copies)]
    state = ops.minmax_state(dev)
    mn = torch.empty(1, dtype=torch.float32, device=dev)
    mx = torch.empty(1, dtype=torch.float32, device=dev)
    ws = torch.empty(max(lib.sbq_stats_workspace_bytes(1, 1, n), 16), dtype=torch.uint8, device=dev)
    t_acc = timed(lambda i: lib.sbq_minmax_accumulate(L.ptr(xs[i % copies]), L.F32, n, L.ptr(state), st))
    t_two = timed(lambda i: lib.sbq_channel_stats(L.ptr(xs[i % copies]), L.F32, 1, 1, n, L.ptr(mn), L.ptr(mx), None, L.ptr(ws), ws.numel(), st))
    line = "%-20s fp32 per tensor: accumulate %.2f us (%.2f TB/s), stats+fold %.2f us" % (name, t_acc, n * 4 / t_acc / 1e6, t_two)
    if len(shape) == 2:
        C, inner = shape
        mnc = torch.empty(C, dtype=torch.float32, device=dev)
        mxc = torch.empty(C, dtype=torch.float32, device=dev)
        t_ch = timed(lambda i: lib.sbq_channel_stats(L.ptr(xs[i % copies]), L.F32, 1, C, inner, L.ptr(mnc), L.ptr(mxc), None, L.ptr(ws), ws.numel(), st))
        line += ", per channel %.2f us (%.2f TB/s)" % (t_ch, n * 4 / t_ch / 1e6)
    print(line)
    xb = [x.bfloat16() for x in xs[:min(copies, 12)]]
    t_b = timed(lambda i: lib.sbq_minmax_accumulate(L.ptr(xb[i % len(xb)]), L.BF16, n, L.ptr(state), st))
    print("%-20s bf16 per tensor: accumulate %.2f us (%.2f TB/s)" % (name, t_b, n * 2 / t_b / 1e6))
    del xs, xb

# (b) per-tensor MSE of 16.7 M bf16
w = (torch.randn(4096, 4096, generator=g) * torch.logspace(-2, 1, 4096).unsqueeze(1)).bfloat16()
xs = [torch.roll(w, i, 1).contiguous().to(dev) for i in range(12)]
n = w.numel()
mn1, mx1, _ = ops.channel_stats(xs[0], 0, False)
sse = torch.zeros(1, 80, dtype=torch.float64, device=dev)
ws = torch.empty(max(lib.sbq_mse_workspace_bytes(1, 1, n), 16), dtype=torch.uint8, device=dev)
for knob in (0, 19):
    L.set_tuning(2, knob)
    t = timed(lambda i: lib.sbq_mse_accumulate(L.ptr(xs[i % 12]), L.BF16, 1, 1, n, L.ptr(mn1), L.ptr(mx1), -128, 127, 1, L.ptr(sse), L.ptr(ws), ws.numel(), st), 20)
    print("per-tensor MSE of 16.7 M bf16, knob2=%d: %.2f us" % (knob, t))
L.set_tuning(2, 0)
