"""Round 6: the batched GPTQ mat-mul on the fp32 matrix cores for 3- and 2-bit levels (gptq_mfma_kernel<MT, BITS>) against
the oracle (cuda_kernel_3bit.cu:85-199, cuda_kernel_2bit.cu:86-153 restated) and against the strip passes it replaces
(knob 2 = 26), HBM-cold.  Literal rtol = atol = 1e-5 (test_cuda_kernel.py:47)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import oracle as O  # noqa: E402
from sparsebit_amd import lib as L  # noqa: E402

dev = torch.device("cuda:0")
lib = L.load()
st = L.stream_ptr(dev)
FN = {4: lib.sbq_vecquant4matmul, 3: lib.sbq_vecquant3matmul, 2: lib.sbq_vecquant2matmul}


def run(bits, in_f, out_f, gs, B, knob2, check):
    g = torch.Generator().manual_seed(bits * 7 + in_f + out_f + B)
    groups = in_f // gs if gs else 1
    rows = in_f // 32 * 3 if bits == 3 else in_f * bits // 32
    copies = max(2, int(3.2e8 // (rows * out_f * 4)) + 1) if in_f * out_f > (1 << 22) else 2
    sets = []
    for _ in range(copies):
        qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (rows, out_f), generator=g, dtype=torch.int64).to(torch.int32)
        sc = (torch.rand(out_f, groups, generator=g) * 0.02 + 0.001).float()
        zr = (torch.randint(0, 2 ** bits, (out_f, groups), generator=g).float() * sc).float()
        sets.append((qw.to(dev), sc.to(dev), zr.to(dev)))
    x = torch.randn(B, in_f, generator=g).float()
    xd = x.to(dev)
    y = torch.zeros(B, out_f, dtype=torch.float32, device=dev)
    ws = L.fresh_workspace(max(lib.sbq_gptq_workspace_bytes(B, in_f, out_f), 16), dev)
    L.set_tuning(2, knob2)

    def call(i):
        qw, sc, zr = sets[i % copies]
        return FN[bits](L.ptr(xd), L.ptr(qw), L.ptr(y), L.ptr(sc), L.ptr(zr), B, in_f, out_f, gs, L.ptr(ws), ws.numel(), st)

    for i in range(5):
        L.check(call(i))
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(50):
        call(i)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / 50
    ok, err = None, float("nan")
    if check:
        y.zero_()
        L.check(call(0))
        torch.cuda.synchronize()
        qw, sc, zr = sets[0]
        ref = O.vecquantmatmul(x.numpy(), qw.cpu().numpy(), np.zeros(out_f, np.float32), sc.cpu().numpy(), zr.cpu().numpy(), gs, bits)
        got = y.cpu().numpy()
        ok = bool(np.all(np.abs(got - ref) <= 1e-5 + 1e-5 * np.abs(ref)))
        err = float(np.abs(got - ref).max())
    L.set_tuning(2, 0)
    return us, ok, err


print("%-4s %-12s %4s %3s  %-6s %-10s %9s %9s" % ("bits", "shape", "gs", "B", "parity", "max_err", "mfma_us", "strips_us"))
bad = 0
for bits in (3, 2, 4):
    for in_f, out_f, gs in ((128, 64, 128), (256, 192, 128), (1024, 4096, 0), (2048, 1024, 256), (4096, 4096, 128), (4096, 11008, 128), (11008, 4096, 128)):
        if in_f % 128:
            continue
        for B in (5, 8, 17, 32):
            us, ok, err = run(bits, in_f, out_f, gs, B, 0, True)
            us_old = run(bits, in_f, out_f, gs, B, 26, False)[0] if in_f >= 4096 else float("nan")
            bad += 0 if ok else 1
            print("%-4d %5dx%-6d %4d %3d  %-6s %-10.3e %9.2f %9.2f" % (bits, in_f, out_f, gs, B, ok, err, us, us_old), flush=True)
print("ALL OK" if bad == 0 else "%d FAILED" % bad)
