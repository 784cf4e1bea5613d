"""Dev tool: the fp32-tensor paths (what a reference user's fp32 model hits): forward, backward, stats, grouped."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sparsebit_amd import lib as L, ops
lib = L.load(); dev = torch.device("cuda:0"); st = L.stream_ptr(dev)
R = C = 4096; NB = 6
def timed(fn, n=200):
    for i in range(50): fn(i)
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n
scale = torch.full((R,), 0.03, device=dev); zp = torch.zeros(R, device=dev)
for dt, did in ((torch.bfloat16, L.BF16), (torch.float32, L.F32)):
    xs = [torch.randn(R, C, device=dev).to(dt) for _ in range(NB)]
    gys = [torch.randn(R, C, device=dev).to(dt) for _ in range(NB)]
    yf = [torch.empty(R, C, device=dev) for _ in range(NB)]
    gx = [torch.empty(R, C, device=dev, dtype=dt) for _ in range(NB)]
    esz = xs[0].element_size()
    t = timed(lambda i: lib.sbq_quant_perchannel_forward(L.ptr(xs[i % NB]), did, L.ptr(yf[i % NB]), L.F32, None, 0, L.ptr(scale), L.ptr(zp), 1, R, C, -128, 127, 0, st))
    print("%-8s forward -> fp32     : %6.2f us  %.2f TB/s" % (str(dt)[6:], t, R * C * (esz + 4) / t / 1e6))
    ws = torch.empty(max(lib.sbq_backward_workspace_bytes(1, R, C), 16), dtype=torch.uint8, device=dev)
    gs = torch.empty(R, device=dev); gz = torch.empty(R, device=dev)
    t = timed(lambda i: lib.sbq_quant_perchannel_backward(L.ptr(xs[i % NB]), L.ptr(gys[i % NB]), did, L.ptr(gx[i % NB]), did, L.ptr(gs), L.ptr(gz), L.ptr(scale), L.ptr(zp), 1, R, C, -128, 127, 0, L.ptr(ws), ws.numel(), st))
    print("%-8s backward gx+gs+gzp  : %6.2f us  %.2f TB/s" % (str(dt)[6:], t, R * C * 3 * esz / t / 1e6))
    mn = torch.empty(R, device=dev); mx = torch.empty(R, device=dev)
    ws2 = torch.empty(max(lib.sbq_stats_workspace_bytes(1, R, C), 16), dtype=torch.uint8, device=dev)
    t = timed(lambda i: lib.sbq_channel_stats(L.ptr(xs[i % NB]), did, 1, R, C, L.ptr(mn), L.ptr(mx), None, L.ptr(ws2), ws2.numel(), st))
    print("%-8s min/max stats       : %6.2f us  %.2f TB/s" % (str(dt)[6:], t, R * C * esz / t / 1e6))
    ws3 = torch.empty(max(lib.sbq_stats_workspace_bytes(1, 1, R * C), 16), dtype=torch.uint8, device=dev)
    t = timed(lambda i: lib.sbq_channel_stats(L.ptr(xs[i % NB]), did, 1, 1, R * C, L.ptr(mn), L.ptr(mx), None, L.ptr(ws3), ws3.numel(), st))
    print("%-8s min/max per tensor  : %6.2f us  %.2f TB/s (two kernels)" % (str(dt)[6:], t, R * C * esz / t / 1e6))
# grouped fp32 weights (ResNet-50-like)
shapes = []
inp = 64
for width, blocks in ((64, 3), (128, 4), (256, 6), (512, 3)):
    for b in range(blocks):
        shapes += [(width, inp, 1, 1), (width, width, 3, 3), (width * 4, width, 1, 1)]
        if b == 0: shapes.append((width * 4, inp, 1, 1))
        inp = width * 4
shapes.append((1000, 2048))
ws_ = [torch.randn(s, device=dev) for s in shapes]
entries = [(w, torch.full((w.shape[0],), 0.05, device=dev), torch.zeros(w.shape[0], device=dev), -8, 7) for w in ws_]
n_el = sum(w.numel() for w in ws_)
gq = ops.GroupFakeQuant(entries)
t = timed(lambda i: gq(), 100)
print("grouped fp32 forward (53 weights, %.1f M): %6.2f us  %.2f TB/s" % (n_el / 1e6, t, n_el * 8 / t / 1e6))
gb = ops.GroupFakeQuantBackward(entries, lsq=True, want_gs=True, gs_ratios=[0.1] * len(entries))
gys = [torch.randn_like(w) for w in ws_]
t = timed(lambda i: gb(gys), 100)
print("grouped fp32 backward                    : %6.2f us  %.2f TB/s" % (t, n_el * 12 / t / 1e6))
