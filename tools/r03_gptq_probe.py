"""Dev tool (round 3): GPTQ 4-bit g128 B=1 mat-vec at the LLaMA-7B shapes, HBM-cold (rotating weight copies), over the
K-split override (knob 1 < 128) and the decode / lane-width switches (knob 2)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sparsebit_amd import lib as L
lib = L.load()
dev = torch.device("cuda:0")
st = L.stream_ptr(dev)
stream = torch.cuda.current_stream(dev)

def timed(fn, iters=300, warm=30, rounds=3):
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

g = torch.Generator().manual_seed(1)
for in_f, out_f in ((4096, 4096), (4096, 11008), (11008, 4096), (4096, 12288), (4096, 22016)):
    groups = in_f // 128
    wb = in_f // 8 * out_f * 4
    copies = max(2, int(3.2e8 // wb) + 1)
    qws = [torch.randint(-2**31, 2**31 - 1, (in_f // 8, out_f), generator=g, dtype=torch.int64).to(torch.int32).to(dev) for _ in range(copies)]
    sc = (torch.rand(out_f, groups, generator=g) * 0.02 + 0.001).to(dev)
    zr = (torch.rand(out_f, groups, generator=g) * 0.1).to(dev)
    x = torch.randn(1, in_f, generator=g).to(dev)
    y = torch.zeros(1, out_f, device=dev)
    ws = torch.zeros(lib.sbq_gptq_workspace_bytes(1, in_f, out_f), dtype=torch.uint8, device=dev)
    nbytes = wb + 2 * out_f * groups * 4 + (in_f + 2 * out_f) * 4
    line = "%5d -> %5d (%5.1f MB, %2d copies):" % (in_f, out_f, nbytes / 1e6, copies)
    for k1, k2 in ((0, 0), (1, 0), (2, 0), (3, 0), (4, 0), (6, 0), (8, 0), (0, 1), (1, 1), (0, 6)):
        L.set_tuning(1, k1); L.set_tuning(2, k2)
        def run(i):
            lib.sbq_vecquant4matmul(L.ptr(x), L.ptr(qws[i % copies]), L.ptr(y), L.ptr(sc), L.ptr(zr), 1, in_f, out_f, 128, L.ptr(ws), ws.numel(), st)
        t = timed(run)
        line += "  k1=%d,k2=%d: %5.2f us (%.2f)" % (k1, k2, t, nbytes / t / 8e6)
    L.set_tuning(1, 0); L.set_tuning(2, 0)
    print(line, flush=True)
    del qws
