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
for bits in (4, 3, 2):
  fn = {4: lib.sbq_vecquant4matmul, 3: lib.sbq_vecquant3matmul, 2: lib.sbq_vecquant2matmul}[bits]
  for (M, N) in ((4096, 4096), (4096, 11008), (11008, 4096)):
    for B in (1, 2, 3, 4, 8):
        gs = 128
        groups = M // gs if gs else 1
        rows = M // 32 * 3 if bits == 3 else M * bits // 32
        qw = torch.randint(-2**31, 2**31 - 1, (rows, N), dtype=torch.int32, device=dev)
        sc = torch.rand(N * groups, device=dev) * 0.01; zr = torch.rand(N * groups, device=dev) * 0.1
        x = torch.randn(B, M, device=dev); y = torch.zeros(B, N, device=dev)
        ws = torch.zeros(max(lib.sbq_gptq_workspace_bytes(B, M, N), 16), dtype=torch.uint8, device=dev)
        st = L.stream_ptr(dev)
        def run(i):
            fn(L.ptr(x), L.ptr(qw), L.ptr(y), L.ptr(sc), L.ptr(zr), B, M, N, gs, L.ptr(ws), ws.numel(), st)
        for mode in (((0, 4) if bits != 3 else (0,)) if B <= 2 else ((0, 9) if B <= 4 else (0,))):
            lib.sbq_set_tuning(2, mode)
            t = timed(run)
            nbytes = rows * N * 4 + 2 * N * groups * 4
            label = {0: "single launch", 4: "byte decode", 9: "k-split+fold"}[mode] if B <= 4 else "k-split+fold"
            print("%d-bit B=%d in=%5d out=%5d group=%3d %-12s: %.2f us  (%.2f TB/s on %.1f MB)" % (bits, B, M, N, gs, label, t, nbytes / t / 1e6, nbytes / 1e6), flush=True)
lib.sbq_set_tuning(2, 0)
