"""Dev tool: GPTQ mat-vec kernel time vs group layout (are the strided scale gathers the cost?)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sparsebit_amd import ops, lib as L
dev = torch.device("cuda:0")
lib = L.load()
def timed(fn, iters=200, warm=20):
    for i in range(warm): fn(i)
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters
for (M, N) in ((4096, 4096), (4096, 11008), (11008, 4096)):
    for gs in (128, 0):
        groups = M // gs if gs else 1
        qw = torch.randint(-2**31, 2**31 - 1, (M // 8, N), dtype=torch.int32, device=dev)
        sc = torch.rand(N * groups, device=dev) * 0.01; zr = torch.rand(N * groups, device=dev) * 0.1
        x = torch.randn(1, M, device=dev); y = torch.zeros(1, N, device=dev)
        ws = torch.empty(max(lib.sbq_gptq_workspace_bytes(1, M, N), 16), dtype=torch.uint8, device=dev)
        st = L.stream_ptr(dev)
        def run(i):
            lib.sbq_vecquant4matmul(L.ptr(x), L.ptr(qw), L.ptr(y), L.ptr(sc), L.ptr(zr), 1, M, N, gs, L.ptr(ws), ws.numel(), st)
        for force_old in (0, 9):
            lib.sbq_set_tuning(2, force_old)
            t = timed(run)
            print("in=%5d out=%5d group=%3d %s: %.2f us  (%.2f TB/s on %.1f MB)" % (M, N, gs, "k-split+fold" if force_old else "strip       ", t, (M * N / 2 + 2 * N * groups * 4) / t / 1e6, (M * N / 2 + 2 * N * groups * 4) / 1e6), flush=True)
lib.sbq_set_tuning(2, 0)
