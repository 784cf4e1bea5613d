"""Dev tool (round 6): per-workgroup timeline of the one-launch percentile of four DeiT batches (fp32 or bf16).
  python tools/lab/build_stamps.py && SBQ_LIB=tools/lab/libsbq_stamps.so python tools/lab/r06_pct_stamps.py [bf16]"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from sparsebit_amd import lib as L  # noqa: E402

if os.environ.get("SBQ_LIB"):
    L.LIB_PATH = os.path.abspath(os.environ["SBQ_LIB"])
from sparsebit_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
acts = [torch.randn(64 * 197 * 384, generator=g).to(dev) for _ in range(4)]
if len(sys.argv) > 1 and sys.argv[1] == "bf16":
    acts = [a.bfloat16() for a in acts]
for _ in range(10):
    ops.percentile_select(acts, 1e-3, 0, False)
torch.cuda.synchronize()
L.set_tuning(1, 779)
for _ in range(3):
    ops.percentile_select(acts, 1e-3, 0, False)
torch.cuda.synchronize()
L.set_tuning(1, 0)
key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
sw = ops._select_workspaces[key]
OLD = 256 + 64 * 128 + 8 * 2 * 2048 * 4 + 256
ONE = 256 + 64 * 128 + 8 * 2 * 2048 * 4
s = sw[OLD + ONE: OLD + ONE + 256 * 256].view(torch.int64).reshape(256, 32).cpu().numpy().astype(np.int64)
t0 = s[:, 0].min()
names = {0: "start", 27: "w15:start", 11: "smp_req", 19: "w15:smp_req", 12: "sample barrier", 13: "slabs_req", 8: "plan8", 9: "plan9", 10: "plan10", 2: "plan",
         15: "sweep loop", 3: "swept", 4: "arrived", 5: "last:begin", 16: "adv gathered", 17: "adv placed", 6: "last:adv", 20: "alone1", 24: "r2:flushed", 25: "r3:flushed", 26: "r4:flushed", 7: "end"}
for j, nm in names.items():
    col = s[:, j]
    ok = col >= t0
    if ok.any():
        v = (col[ok] - t0) / 100.0
        print("%-16s n=%4d  min %6.1f  median %6.1f  p90 %6.1f  max %6.1f us" % (nm, ok.sum(), v.min(), np.median(v), np.percentile(v, 90), v.max()))
