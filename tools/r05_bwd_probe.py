"""Round 5: the 16-bit STE / LSQ backward on the headline weight, resident schedule (default) vs the chunked kernel
(knob 3 = 1), HBM-cold (12 rotating x / gy / gx triples = 1.2 GB).   python tools/r05_bwd_probe.py"""
import ctypes
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sparsebit_amd import lib as L  # noqa: E402

dev = torch.device("cuda:0")
lib = L.load()
st = L.stream_ptr(dev)
stream = torch.cuda.current_stream(dev)
R = Cc = 4096
nb = 12
g = torch.Generator().manual_seed(1)
w = (torch.randn(R, Cc, generator=g) * torch.logspace(-2, 1, R).unsqueeze(1)).bfloat16().to(dev)
xs = [torch.roll(w, j, 1).contiguous() for j in range(nb)]
gy = torch.randn(R, Cc, generator=g).bfloat16().to(dev)
gys = [torch.roll(gy, j, 1).contiguous() for j in range(nb)]
gxs = [torch.empty_like(w) for _ in range(nb)]
s_raw = -(w.float().abs().mean(1) * 2 / math.sqrt(7)).contiguous()
z_raw = torch.zeros(R, device=dev)
gs = torch.empty(R, device=dev)
gz = torch.empty(R, device=dev)
ws = torch.empty(max(lib.sbq_backward_workspace_bytes(1, R, Cc), 16), dtype=torch.uint8, device=dev)
ratio = 1.0 / math.sqrt(Cc * 7)


def lsq(i):
    j = i % nb
    return lib.sbq_quant_lsq_backward(L.ptr(xs[j]), L.ptr(gys[j]), L.BF16, L.ptr(gxs[j]), L.BF16, L.ptr(gs), L.ptr(s_raw), L.ptr(z_raw),
                                      1, R, Cc, -8, 7, ctypes.c_float(ratio), L.ptr(ws), ws.numel(), st)


def ste(i):
    j = i % nb
    return lib.sbq_quant_perchannel_backward(L.ptr(xs[j]), L.ptr(gys[j]), L.BF16, L.ptr(gxs[j]), L.BF16, L.ptr(gs), L.ptr(gz),
                                             L.ptr(s_raw.abs()), L.ptr(z_raw), 1, R, Cc, -8, 7, 0, L.ptr(ws), ws.numel(), st)


def gx_only(i):
    j = i % nb
    return lib.sbq_quant_perchannel_backward(L.ptr(xs[j]), L.ptr(gys[j]), L.BF16, L.ptr(gxs[j]), L.BF16, None, None,
                                             L.ptr(s_raw.abs()), L.ptr(z_raw), 1, R, Cc, -8, 7, 0, L.ptr(ws), ws.numel(), st)


def timed(fn, iters=200):
    best = 1e9
    for _ in range(3):
        for i in range(20):
            L.check(fn(i))
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for i in range(iters):
            fn(i)
        b.record(stream)
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / iters)
    return best


nbytes = R * Cc * 6
for name, fn in (("LSQ backward (gx + gs)", lsq), ("STE backward (gx + gs + gzp)", ste), ("STE backward (gx only)", gx_only)):
    out = {}
    for knob in (0, 1):
        L.set_tuning(3, knob)
        try:
            out[knob] = timed(fn)
        finally:
            L.set_tuning(3, 0)
    print("%-30s resident %.2f us (%.3f of 8 TB/s)   chunked %.2f us (%.3f)" % (name, out[0], nbytes / out[0] / 1e3 / 8000, out[1],
                                                                               nbytes / out[1] / 1e3 / 8000), flush=True)
