"""Round 5: the batched GPTQ mat-mul (gptq_mfma_kernel, 5 <= B <= 32) against the oracle and against the strip tiles of
four rows it replaces (knob 2 = 26), HBM-cold (weight copies in rotation > 256 MiB).

    python tools/r05_gptq_batch.py > profiles/r05_gptq_batch.log
"""
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
stream = torch.cuda.current_stream(dev)


def problem(in_f, out_f, gs, seed, copies):
    g = torch.Generator().manual_seed(seed)
    groups = in_f // gs if gs else 1
    out = []
    for _ in range(copies):
        qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (in_f // 8, out_f), generator=g, dtype=torch.int64).to(torch.int32)
        sc = (torch.rand(out_f, groups, generator=g) * 0.02 + 0.001).float()
        zr = (torch.randint(0, 16, (out_f, groups), generator=g).float() * sc).float()
        out.append((qw.to(dev), sc.to(dev), zr.to(dev)))
    return out, g


def timed(fn, iters=200, warm=20):
    best = 1e9
    for _ in range(2):
        for i in range(warm):
            fn(i)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for i in range(iters):
            fn(i)
        b.record(stream)
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / iters)
    return best


K1 = int(os.environ.get("KNOB1", "0"))
L.set_tuning(1, K1)
ONLY_TIMED = os.environ.get("ONLY_TIMED") == "1"
print("knob 1 (blocks per wave) = %d" % K1)
print("shape            gs   B   parity  max_err     mfma_us  strip4_us  frac_fp32_peak  frac_hbm")
ok_all = True
for in_f, out_f, gs, time_it in ((4096, 4096, 128, True), (4096, 11008, 128, True), (11008, 4096, 128, True), (128, 64, 128, False),
                                 (256, 192, 128, False), (1024, 4096, 0, False), (2048, 1024, 256, False), (12288, 4096, 128, True)):
    if ONLY_TIMED and not time_it:
        continue
    w_bytes = in_f // 8 * out_f * 4
    copies = max(2, int(3.2e8 // w_bytes) + 1) if time_it else 1
    mats, g = problem(in_f, out_f, gs, 7 + in_f % 97, copies)
    for B in ((5, 8, 16, 17, 29, 32) if not time_it else (8, 16, 32)):
        x = torch.randn(B, in_f, generator=g).float()
        xd = x.to(dev)
        bias = torch.randn(out_f, generator=g).float()
        y = bias.repeat(B, 1).to(dev)
        ws = L.fresh_workspace(max(lib.sbq_gptq_workspace_bytes(B, in_f, out_f), 16), dev)

        def run(i):
            m = mats[i % copies]
            return lib.sbq_vecquant4matmul(L.ptr(xd), L.ptr(m[0]), L.ptr(y), L.ptr(m[1]), L.ptr(m[2]), B, in_f, out_f, gs, L.ptr(ws),
                                           ws.numel(), st)

        L.check(run(0))
        torch.cuda.synchronize()
        ref = O.vecquantmatmul(x.numpy(), mats[0][0].cpu().numpy(), bias.numpy(), mats[0][1].cpu().numpy(), mats[0][2].cpu().numpy(),
                               gs, 4)
        got = y.cpu().numpy()
        tol = 1e-5 * max(1.0, float(np.abs(ref).max()))
        ok = bool(np.all(np.abs(got - ref) <= tol + 1e-5 * np.abs(ref)))
        # determinism: a second call adds the same sums again
        y2 = bias.repeat(B, 1).to(dev)
        lib.sbq_vecquant4matmul(L.ptr(xd), L.ptr(mats[0][0]), L.ptr(y2), L.ptr(mats[0][1]), L.ptr(mats[0][2]), B, in_f, out_f, gs,
                                L.ptr(ws), ws.numel(), st)
        torch.cuda.synchronize()
        ok = ok and bool(torch.equal(y2.cpu(), torch.from_numpy(got)))
        ok_all = ok_all and ok
        us = us4 = float("nan")
        if time_it:
            us = timed(run)
            L.set_tuning(2, 26)
            try:
                us4 = timed(run, 100, 10)
            finally:
                L.set_tuning(2, 0)
        flops = 2.0 * B * in_f * out_f
        groups = in_f // gs if gs else 1
        nbytes = w_bytes + 2 * out_f * groups * 4 + B * (in_f + 2 * out_f) * 4
        print("%5dx%-5d %5d %3d   %-5s  %.3e  %8.2f  %8.2f      %.3f          %.3f" % (
            in_f, out_f, gs, B, ok, float(np.abs(got - ref).max()), us, us4, flops / us / 1e6 / 157.3 if time_it else float("nan"),
            nbytes / us / 1e3 / 8000 if time_it else float("nan")), flush=True)
print("ALL OK" if ok_all else "FAILURES")
