"""Dev tool: fused observe + QDQ of a weight vs the separate steps (C ABI loops, rotating buffers)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sparsebit_amd import lib as L
lib = L.load()
dev = torch.device("cuda:0"); st = L.stream_ptr(dev)
def timed(fn, iters=300, warm=30):
    best = 1e9
    for _ in range(3):
        for i in range(warm): fn(i)
        torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(iters): fn(i)
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / iters)
    return best
for rows, inner in ((4096, 4096), (11008, 4096), (4096, 2048)):
    for dt, did in ((torch.bfloat16, 2), (torch.float32, 0)):
        nbuf = 12
        xs = [(torch.randn(rows, inner, device=dev) * torch.logspace(-2, 1, rows, device=dev).unsqueeze(1)).to(dt) for _ in range(nbuf)]
        ys = [torch.empty_like(x) for x in xs]
        st4 = torch.empty(4, rows, device=dev)
        ws = torch.empty(lib.sbq_stats_workspace_bytes(1, rows, inner) + 1024, dtype=torch.uint8, device=dev)
        def fused(i):
            j = i % nbuf
            rc = lib.sbq_observe_quant_perchannel_forward(L.ptr(xs[j]), did, L.ptr(ys[j]), did, L.ptr(st4[0]), L.ptr(st4[1]), L.ptr(st4[2]), L.ptr(st4[3]), rows, inner, -128, 127, 1, L.ptr(ws), ws.numel(), st)
            assert rc == 0
        def three(i):
            j = i % nbuf
            lib.sbq_channel_stats(L.ptr(xs[j]), did, 1, rows, inner, L.ptr(st4[2]), L.ptr(st4[3]), None, L.ptr(ws), ws.numel(), st)
            lib.sbq_qparams_from_minmax(L.ptr(st4[2]), L.ptr(st4[3]), rows, -128, 127, 1, L.ptr(st4[0]), L.ptr(st4[1]), st)
            lib.sbq_quant_perchannel_forward(L.ptr(xs[j]), did, L.ptr(ys[j]), did, None, 0, L.ptr(st4[0]), L.ptr(st4[1]), 1, rows, inner, -128, 127, 0, st)
        tf = timed(fused)
        t3 = timed(three)
        n = rows * inner
        print("%5d x %4d %-8s fused %6.2f us (%.2f TB/s of 2x%dB/elem)   three steps %6.2f us" % (rows, inner, str(dt)[6:], tf, n * 2 * xs[0].element_size() / tf / 1e6, xs[0].element_size(), t3), flush=True)
        del xs, ys
