"""Dev tool: QDQ rate over the weight / activation shapes of SURVEY.md 8(d)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sparsebit_amd import lib as L
lib = L.load()
dev = torch.device("cuda:0")
st = L.stream_ptr(dev)
def timed(fn, iters=100, warm=20):
    for i in range(warm): fn(i)
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters
# warm clocks
w = torch.randn(8192, 8192, device=dev).bfloat16(); o = torch.empty_like(w)
s = torch.ones(8192, device=dev); z = torch.zeros(8192, device=dev)
for _ in range(300):
    lib.sbq_quant_perchannel_forward(L.ptr(w), 2, L.ptr(o), 2, None, 0, L.ptr(s), L.ptr(z), 1, 8192, 8192, -128, 127, 0, st)
torch.cuda.synchronize()
cases = [("4096x4096 w", (4096, 4096), 0, True), ("11008x4096 w", (11008, 4096), 0, True), ("4096x11008 w", (4096, 11008), 0, True),
         ("512x512x3x3 w", (512, 512, 3, 3), 0, True), ("1000x512 w", (1000, 512), 0, True), ("64x3x7x7 w", (64, 3, 7, 7), 0, True),
         ("1536x384 w", (1536, 384), 0, True), ("64x197x384 act/tensor", (64, 197, 384), 2, False),
         ("256x64x56x56 act/tensor", (256, 64, 56, 56), 1, False), ("64x197x1536 act/channel", (64, 197, 1536), 2, True),
         ("32x256x56x56 act/channel", (32, 256, 56, 56), 1, True)]
for name, shape, ch_axis, perch in cases:
    n = 1
    for d in shape: n *= d
    C = shape[ch_axis] if perch else 1
    outer = 1
    for d in shape[:ch_axis]: outer *= d
    inner = n // (outer * C) if perch else n
    if not perch: outer = 1
    sc = (torch.rand(C, device=dev) * 0.05 + 0.01); zp = torch.zeros(C, device=dev)
    line = "%-26s %10d elem  outer=%d C=%d inner=%d :" % (name, n, outer, C, inner)
    for dt, did, esz in ((torch.bfloat16, 2, 2), (torch.float32, 0, 4)):
        nbuf = max(2, min(12, int(6e8 // (n * 2 * esz)) + 1))
        xs = [torch.randn(*shape, device=dev).to(dt) for _ in range(nbuf)]
        ys = [torch.empty_like(x) for x in xs]
        def run(i):
            j = i % nbuf
            rc = lib.sbq_quant_perchannel_forward(L.ptr(xs[j]), did, L.ptr(ys[j]), did, None, 0, L.ptr(sc), L.ptr(zp), outer, C, inner, -128, 127, 0, st)
            assert rc == 0
        t = timed(run)
        line += "  %s %8.2f us %5.2f TB/s" % ("bf16" if esz == 2 else "fp32", t, n * 2 * esz / t / 1e6)
        del xs, ys
    print(line, flush=True)
