"""Dev tool (round 4): the headline launch loop as a hipGraph (torch.cuda.CUDAGraph capturing the C-ABI calls) vs plain
launches: us per step, K = 20 and K = 1024, 12 rotating tensors."""
import os
import sys
import time

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


def run(i, st):
    j = i % 12
    rc = lib.sbq_quant_perchannel_forward(L.ptr(xs[j]), L.BF16, L.ptr(ys[j]), L.BF16, None, L.Q_NONE, L.ptr(scale), L.ptr(zp), 1, R, C,
                                          -128, 127, 0, st)
    assert rc == 0


def timed_plain(K):
    st = L.stream_ptr(dev)
    for i in range(50):
        run(i, st)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for i in range(K):
            run(i, st)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e6 / K)
    return best


def timed_graph(K):
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(12):
            run(i, L.stream_ptr(dev))
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    with torch.cuda.graph(graph):
        st = L.stream_ptr(dev)
        for i in range(K):
            run(i, st)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        graph.replay()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e6 / K)
    return best


for K in (20, 200, 1024):
    print("K = %4d: plain launches %.2f us per step, one graph of K launches %.2f us per step" % (K, timed_plain(K), timed_graph(K)))
