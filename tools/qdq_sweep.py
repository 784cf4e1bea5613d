"""Kernel-development tool: sweep launch variants of the forward QDQ on MI355X.

Usage (GPU box):  python tools/qdq_sweep.py [rows cols]
Prints one line per (variant, grid cap) with HBM-cold (rotating buffers, larger
than the 256 MiB Infinity Cache) and cache-warm timings.  Not part of the product.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sparsebit_amd import lib as L  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    cols = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    dev = torch.device("cuda:0")
    lib = L.load(strict=False)
    nbuf = 12 if rows * cols <= 4096 * 4096 else 3
    g = torch.Generator().manual_seed(0)
    w = torch.randn(rows, cols, generator=g) * torch.logspace(-2, 1, rows).unsqueeze(1)
    xs = [w.bfloat16().to(dev) for _ in range(nbuf)]
    ys = [torch.empty_like(xs[0]) for _ in range(nbuf)]
    xf = xs[0].float()
    mx = xf.abs().amax(1)
    scale = torch.clamp(mx * 2 / 255.0, min=1e-6).contiguous()
    zp = torch.zeros_like(scale)
    st = L.stream_ptr()
    n = rows * cols

    def run(i):
        rc = lib.sbq_quant_perchannel_forward(
            L.ptr(xs[i]), L.BF16, L.ptr(ys[i]), L.BF16, None, L.Q_NONE, L.ptr(scale), L.ptr(zp),
            1, rows, cols, -128, 127, 0, st)
        assert rc == 0, rc

    # sanity vs torch on device (not the parity proof: see tests/)
    run(0)
    torch.cuda.synchronize()
    ref = ((torch.clamp(torch.round(xf / scale[:, None]), -128, 127)) * scale[:, None]).bfloat16()
    bad = (ref != ys[0]).sum().item()
    print("sanity mismatches vs torch-gpu:", bad, "of", n)

    def timeit(fn, iters):
        for i in range(10):
            fn(i % nbuf)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            fn(i % nbuf)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / iters  # us

    def timeit_graph(fn, iters):
        # hipGraph of `iters` launches: removes host launch gaps from the measurement
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            stp = L.stream_ptr()
            for i in range(3):
                fn(i % nbuf, stp)
            torch.cuda.synchronize()
            gph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gph, stream=s):
                stp = L.stream_ptr()
                for i in range(iters):
                    fn(i % nbuf, stp)
        gph.replay()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        gph.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / iters

    bytes_alg = n * 4
    t = timeit(lambda i: ys[i].copy_(xs[i]), 200)
    print("torch copy_ bf16 cold: %.2f us  %.2f TB/s" % (t, bytes_alg / t / 1e6))
    t = timeit(lambda i: ys[0].copy_(xs[0]), 200)
    print("torch copy_ bf16 warm: %.2f us  %.2f TB/s" % (t, bytes_alg / t / 1e6))

    def runp(i, stp):
        rc = lib.sbq_quant_perchannel_forward(
            L.ptr(xs[i]), L.BF16, L.ptr(ys[i]), L.BF16, None, L.Q_NONE, L.ptr(scale), L.ptr(zp),
            1, rows, cols, -128, 127, 0, stp)
        assert rc == 0, rc

    names = {0: "fast ", 3: "ieee "}
    for _ in range(3000):
        run(0)
    torch.cuda.synchronize()
    lib.sbq_set_tuning(2, 0)
    for rep in range(2):
        for variant, label in ((0, "loads nt , stores nt "), (8, "loads nt , stores wb "), (12, "loads wb , stores nt "), (4, "loads wb , stores wb ")):
            lib.sbq_set_tuning(0, variant)
            lib.sbq_set_tuning(1, 0)
            cold = min(timeit(run, 300) for _ in range(3))
            warm = min(timeit(lambda i: run(0), 300) for _ in range(2))
            print("%s : cold %.2f us %.2f TB/s | warm %.2f us" % (label, cold, bytes_alg / cold / 1e6, warm), flush=True)
    lib.sbq_set_tuning(2, 0)
    lib.sbq_set_tuning(0, -1)
    lib.sbq_set_tuning(1, 0)


if __name__ == "__main__":
    main()
