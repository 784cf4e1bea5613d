"""Dev tool: kth_value / percentile on small and odd sizes against torch.sort (single process)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from sparsebit_amd import ops
g = torch.Generator().manual_seed(11)
bad = 0
for dt in (torch.float32, torch.bfloat16, torch.float16):
    for n in (8, 100, 4096, 16384, 24576, 24577, 40000, 100000, 262144, 1 << 20):
        x = torch.randn(n, generator=g).to(dt).cuda()
        for ua in (False, True):
            ref = torch.sort((x.abs() if ua else x).float())[0]
            for k in sorted({1, 2, n // 2, n // 2 + 1, n - 1, n, max(1, n // 1000)}):
                for rep in range(3):
                    got = float(ops.kth_value(x, k, ua))
                    if got != float(ref[k - 1]):
                        bad += 1
                        if bad < 20: print("MISMATCH", dt, n, ua, k, rep, got, float(ref[k - 1]), flush=True)
print("bad", bad)
