"""Dev tool (round 3): A/B timings of the kernels reworked this round through the tuning knobs, C ABI loops,
rotating HBM-resident inputs.  knob 2: 11 = general statistics kernel, 12 = multi-launch selection protocol."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sparsebit_amd import lib as L, ops
lib = L.load()
dev = torch.device("cuda:0")
st = L.stream_ptr(dev)
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

R = C = 4096
NB = 12
g = torch.Generator().manual_seed(0)
w0 = (torch.randn(R, C, generator=g) * torch.logspace(-2, 1, R).unsqueeze(1))
for dt, did in ((torch.bfloat16, L.BF16), (torch.float32, L.F32)):
    xs = [w0.to(dt).to(dev)]
    for i in range(1, NB): xs.append(torch.roll(xs[0], i, 1).contiguous())
    n = R * C
    esz = xs[0].element_size()
    mn = torch.empty(R, dtype=torch.float32, device=dev); mx = torch.empty_like(mn)
    ws = torch.empty(max(lib.sbq_stats_workspace_bytes(1, R, C), 16), dtype=torch.uint8, device=dev)
    for knob in (0, 11):
        L.set_tuning(2, knob)
        t = timed(lambda i: lib.sbq_channel_stats(L.ptr(xs[i % NB]), did, 1, R, C, L.ptr(mn), L.ptr(mx), None, L.ptr(ws), ws.numel(), st))
        tt = timed(lambda i: lib.sbq_channel_stats(L.ptr(xs[i % NB]), did, 1, 1, n, L.ptr(mn), L.ptr(mx), None, L.ptr(ws), ws.numel(), st))
        print("%s stats per-channel knob2=%d: %.2f us (%.2f TB/s)   per-tensor: %.2f us" % (dt, knob, t, n * esz / t / 1e6, tt), flush=True)
    L.set_tuning(2, 0)
    sw = torch.zeros(lib.sbq_radix_select_workspace_bytes(1, 2), dtype=torch.uint8, device=dev)
    out = torch.empty(2, dtype=torch.float32, device=dev)
    import ctypes
    for knob in (0, 12):
        L.set_tuning(2, knob)
        t = timed(lambda i: lib.sbq_kth_value(L.ptr(xs[i % NB]), did, n, 1, n // 2 + 1, L.ptr(out), L.ptr(sw), sw.numel(), st), 100, 10)
        def pct(i):
            p = (ctypes.c_void_p * 1)(xs[i % NB].data_ptr()); o = (ctypes.c_int64 * 1)(1)
            lib.sbq_percentile_select(p, o, 1, did, 1, n, 1e-3, L.ptr(out[0:1]), L.ptr(out[1:2]), L.ptr(sw), sw.numel(), st)
        t2 = timed(pct, 100, 10)
        print("%s kth_value knob2=%d: %.2f us (%.2f TB/s)   percentile per tensor: %.2f us" % (dt, knob, t, n * esz / t / 1e6, t2), flush=True)
    L.set_tuning(2, 0)
    # DeiT-small: 4 cached batches per tensor
    bs = [[torch.randn(64, 197, 384, generator=g).to(dt).to(dev) for _ in range(4)] for _ in range(3)]
    for knob in (0, 12):
        L.set_tuning(2, knob)
        t = timed(lambda i: ops.percentile_select(bs[i % 3], 1e-3, 0, False), 50, 5)
        print("%s DeiT 4 batches percentile knob2=%d: %.2f us" % (dt, knob, t), flush=True)
    L.set_tuning(2, 0)
    del xs, bs
