"""Dev tool (round 4): what the closing torch.cuda.synchronize() costs a K = 20 timed region, blocking vs spinning
(hipSetDeviceFlags(hipDeviceScheduleSpin) before the context exists) vs polling an event."""
import ctypes
import os
import sys
import time

MODE = sys.argv[1] if len(sys.argv) > 1 else "default"
if MODE == "spin":
    hip = ctypes.CDLL("libamdhip64.so")
    print("hipSetDeviceFlags(spin) ->", hip.hipSetDeviceFlags(1))
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from sparsebit_amd import lib as L

lib = L.load()
dev = torch.device("cuda:0")
R = C = 4096
g = torch.Generator().manual_seed(0)
w = (torch.randn(R, C, generator=g) * torch.logspace(-2, 1, R).unsqueeze(1)).bfloat16().to(dev)
xs = [w] + [torch.roll(w, i, 1).contiguous() for i in range(1, 12)]
ys = [torch.empty_like(x) for x in xs]
scale = (w.float().abs().amax(1) / 127).contiguous()
zp = torch.zeros(R, device=dev)
st = L.stream_ptr(dev)


def run(i):
    j = i % 12
    lib.sbq_quant_perchannel_forward(L.ptr(xs[j]), L.BF16, L.ptr(ys[j]), L.BF16, None, L.Q_NONE, L.ptr(scale), L.ptr(zp), 1, R, C, -128, 127, 0, st)


for i in range(200):
    run(i)
torch.cuda.synchronize()
for K in (20, 200):
    res = []
    for _ in range(9):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            run(i)
        if MODE == "poll":
            ev = torch.cuda.Event()
            ev.record()
            while not ev.query():
                pass
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) * 1e6 / K)
    res.sort()
    print("%s K = %3d: %.2f us per step (median of 9; min %.2f)" % (MODE, K, res[4], res[0]))
