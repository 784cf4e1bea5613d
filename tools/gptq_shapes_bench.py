"""Dev tool: GPTQ 4-bit g128 mat-vec over the reference's KAT shapes (test_cuda_kernel.py:50-126) and the LLaMA-7B
linears: time per call (events, back-to-back on one stream) and weight-stream rate vs the 8 TB/s HBM peak."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sparsebit_amd import lib as L, ops
def timed(fn, iters=50, warm=10):
    best = 1e9
    for _ in range(3):
        for _ in range(warm): fn()
        torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters): fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / iters)
    return best
shapes = [(4096, 4096), (4096, 11008), (11008, 4096), (8192, 8192), (8192, 32768), (9216, 36864), (12288, 49152)]
batches = [1, 4, 8, 32]
_b = batches
bits = int(sys.argv[1]) if len(sys.argv) > 1 else 4
if len(sys.argv) > 2:
    L.set_tuning(2, int(sys.argv[2]))  # 1 / 2: 128 / 64 channels per K lane, 9: two-launch path
if len(sys.argv) > 3:
    batches = [int(a) for a in sys.argv[3].split(",")]
if len(sys.argv) > 4:
    L.set_tuning(1, int(sys.argv[4]))  # K split override (dev)
if len(sys.argv) > 5:
    shapes = shapes[int(sys.argv[5]):]
print("bits %d group 128; bytes = qweight + scales + zeros + x + 2 * out" % bits)
for in_f, out_f in shapes:
    g = torch.Generator().manual_seed(1)
    rows = (in_f + 31) // 32 * 3 if bits == 3 else in_f * bits // 32
    # several copies of the weights so that back-to-back calls do not hit the 256 MiB Infinity Cache
    nbytes = rows * out_f * 4
    ncopy = max(1, min(8, int(6e8 // nbytes)))
    qws = [torch.randint(-2 ** 31, 2 ** 31 - 1, (rows, out_f), generator=g, dtype=torch.int64).to(torch.int32).cuda() for _ in range(ncopy)]
    groups = in_f // 128
    scales = (torch.rand(out_f, groups, generator=g) * 0.02 + 0.001).cuda()
    zeros = (scales.cpu() * torch.randint(0, 2 ** bits, (out_f, groups), generator=g).float()).cuda()
    line = "%6d x %6d (%6.1f MB x %d):" % (in_f, out_f, nbytes / 1e6, ncopy)
    for b in batches:
        x = torch.randn(b, in_f, generator=g).cuda()
        out = torch.zeros(b, out_f, device="cuda")
        i = [0]
        def run():
            ops.vecquantmatmul(bits, x, qws[i[0] % ncopy], out, scales, zeros, 128)
            i[0] += 1
        t = timed(run)
        total = nbytes + 2 * out_f * groups * 4 + b * in_f * 4 + 2 * b * out_f * 4
        line += "  B=%-2d %7.1f us %5.2f TB/s (%.2f)" % (b, t, total / t / 1e6, total / t / 1e6 / 8.0)
    print(line, flush=True)
    del qws
    torch.cuda.empty_cache()
