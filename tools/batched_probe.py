"""Dev tool: why is the batched launch slower per weight than one big tensor?"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sparsebit_amd import ops, lib as L
dev = torch.device("cuda:0")
R = C = 4096
n = R * C
def timed(fn, iters=60, warm=10):
    for i in range(warm): fn(i)
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters
scale = (torch.rand(R, device=dev) * 0.05 + 0.01); zp = torch.zeros(R, device=dev)
# warm the clocks
big = [torch.randn(12 * R, C, device=dev).bfloat16() for _ in range(2)]
bigo = [torch.empty_like(b) for b in big]
sc12 = scale.repeat(12).contiguous(); zp12 = zp.repeat(12).contiguous()
for _ in range(3):
    t = timed(lambda i: ops.fake_quant(big[i % 2], sc12, zp12, -128, 127, 0, out_dtype=torch.bfloat16), 30)
print("one 49152x4096 tensor, single-tensor kernel : %.2f us per 4096x4096" % (t / 12))
for nit in (1, 2, 6, 12):
    # (a) items = views into the big contiguous tensors
    groups = []
    for gi in range(2):
        xs = [big[gi][k * R:(k + 1) * R] for k in range(nit)]
        ys = [bigo[gi][k * R:(k + 1) * R] for k in range(nit)]
        groups.append(ops.BatchedFakeQuant(xs, [scale] * nit, [zp] * nit, -128, 127, 0, torch.bfloat16, outs=ys))
    t = timed(lambda i: groups[i % 2](), 100 if nit < 6 else 40)
    print("batched n=%2d, views of one allocation        : %.2f us per weight" % (nit, t / nit))
sep = [[torch.randn(R, C, device=dev).bfloat16() for _ in range(6)] for _ in range(2)]
sepo = [[torch.empty_like(x) for x in g] for g in sep]
groups = [ops.BatchedFakeQuant(sep[g], [scale] * 6, [zp] * 6, -128, 127, 0, torch.bfloat16, outs=sepo[g]) for g in range(2)]
t = timed(lambda i: groups[i % 2](), 40)
print("batched n= 6, separate allocations            : %.2f us per weight" % (t / 6))
lib = L.load()
for cap in (512, 768, 2048):
    lib.sbq_set_tuning(1, cap)
    t = timed(lambda i: groups[i % 2](), 40)
    print("batched n= 6, separate allocations, grid %4d : %.2f us per weight" % (cap, t / 6))
lib.sbq_set_tuning(1, 0)
