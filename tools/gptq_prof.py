"""Dev tool for rocprofv3: the GPTQ 4-bit g128 mat-vec at the reference's KAT sizes, a few launches each."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sparsebit_amd import ops, lib as L
if len(sys.argv) > 1:
    L.set_tuning(2, int(sys.argv[1]))  # e.g. 5: (strip, K block) grid instead of the persistent workers
for in_f, out_f in ((4096, 4096), (8192, 32768), (12288, 49152)):
    g = torch.Generator().manual_seed(1)
    rows = in_f * 4 // 32
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (rows, out_f), generator=g, dtype=torch.int64).to(torch.int32).cuda()
    groups = in_f // 128
    scales = (torch.rand(out_f, groups, generator=g) * 0.02 + 0.001).cuda()
    zeros = (scales.cpu() * torch.randint(0, 16, (out_f, groups), generator=g).float()).cuda()
    for b in (1, 4):
        x = torch.randn(b, in_f, generator=g).cuda()
        out = torch.zeros(b, out_f, device="cuda")
        for _ in range(12):
            ops.vecquantmatmul(4, x, qw, out, scales, zeros, 128)
        torch.cuda.synchronize()
    del qw
    torch.cuda.empty_cache()
