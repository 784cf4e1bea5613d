"""Dev tool: launch the headline QDQ N times (rotating buffers) for rocprofv3."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sparsebit_amd import lib as L

variant = int(sys.argv[1]) if len(sys.argv) > 1 else -1
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 0
math = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lib = L.load(strict=False)
lib.sbq_set_tuning(0, variant); lib.sbq_set_tuning(1, cap); lib.sbq_set_tuning(2, math)
dev = torch.device("cuda:0")
rows = cols = 4096
nbuf = 12
g = torch.Generator().manual_seed(0)
w = torch.randn(rows, cols, generator=g) * torch.logspace(-2, 1, rows).unsqueeze(1)
xs = [w.bfloat16().to(dev) for _ in range(nbuf)]
ys = [torch.empty_like(xs[0]) for _ in range(nbuf)]
scale = torch.clamp(xs[0].float().abs().amax(1) * 2 / 255.0, min=1e-6).contiguous()
zp = torch.zeros_like(scale)
st = L.stream_ptr()
for i in range(200):
    j = i % nbuf
    rc = lib.sbq_quant_perchannel_forward(L.ptr(xs[j]), L.BF16, L.ptr(ys[j]), L.BF16, None, L.Q_NONE,
                                          L.ptr(scale), L.ptr(zp), 1, rows, cols, -128, 127, 0, st)
    assert rc == 0
torch.cuda.synchronize()
