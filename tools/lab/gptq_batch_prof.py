"""Driver for rocprofv3 (tools/r05_gptq_batch_pmc.sh): the batched GPTQ mat-mul (gptq_mfma_kernel), 4-bit g128, the three
LLaMA-7B shapes at B = 8 and B = 32, HBM-cold (weight copies in rotation), 120 launches each."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from sparsebit_amd import lib as L  # noqa: E402

dev = torch.device("cuda:0")
lib = L.load()
st = L.stream_ptr(dev)
g = torch.Generator().manual_seed(1)
for in_f, out_f in ((4096, 4096), (4096, 11008), (11008, 4096)):
    groups = in_f // 128
    wb = in_f // 8 * out_f * 4
    copies = max(2, int(3.2e8 // wb) + 1)
    qws = [torch.randint(-2 ** 31, 2 ** 31 - 1, (in_f // 8, out_f), generator=g, dtype=torch.int64).to(torch.int32).to(dev)
           for _ in range(copies)]
    sc = (torch.rand(out_f, groups, generator=g) * 0.02 + 0.001).to(dev)
    zr = (torch.rand(out_f, groups, generator=g) * 0.1).to(dev)
    for B in (8, 32):
        x = torch.randn(B, in_f, generator=g).to(dev)
        y = torch.zeros(B, out_f, device=dev)
        ws = L.fresh_workspace(lib.sbq_gptq_workspace_bytes(B, in_f, out_f), dev)
        for i in range(120):
            L.check(lib.sbq_vecquant4matmul(L.ptr(x), L.ptr(qws[i % copies]), L.ptr(y), L.ptr(sc), L.ptr(zr), B, in_f, out_f, 128,
                                            L.ptr(ws), ws.numel(), st))
        torch.cuda.synchronize()
    del qws
