"""Dev tool (round 6): per-workgroup timeline of the grouped fp32 selection (group_kth_kernel) on ResNet-50's 53 weights.
  python tools/lab/build_stamps.py && SBQ_LIB=tools/lab/libsbq_stamps.so python tools/lab/r06_group_stamps.py
Stamps (s_memrealtime, 100 MHz) per workgroup: 0 start, 12 sample in, 2 plan done, 3 sweep done, 4 arrived, 5 last arriver
starts, 6 advance done, 20.. the last arriver's sweeps alone, 7 end."""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from sparsebit_amd import lib as L  # noqa: E402

if os.environ.get("SBQ_LIB"):
    L.LIB_PATH = os.environ["SBQ_LIB"]
import bench_configs as B  # noqa: E402
from sparsebit_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(50)
ws = [torch.randn(s, generator=g).to(dev) for s in B.resnet50_weight_shapes()]
ks = [min(int(w.numel() * 0.5), w.numel() - 1) + 1 for w in ws]
if os.environ.get("SBQ_ONLY"):  # one tensor of the list only
    ws, ks = [ws[int(os.environ["SBQ_ONLY"])]], [ks[int(os.environ["SBQ_ONLY"])]]
    print("only tensor: %d elements" % ws[0].numel())
if os.environ.get("SBQ_KNOB2"):
    L.set_tuning(2, int(os.environ["SBQ_KNOB2"]))
for _ in range(5):
    ops.group_kth_value(ws, ks, True)
torch.cuda.synchronize()
key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
wsb = ops._group_kth_workspaces[key]
nb = 2048 * 32 * 8
total = wsb.numel()
wsb[-nb:].zero_()
ops.group_kth_value(ws, ks, True)
torch.cuda.synchronize()
st = wsb[-nb:].cpu().numpy().view(np.uint64).reshape(2048, 32).astype(np.int64)
used = st[:, 0] > 0
anyst = (st != 0).any(axis=1)
print("rows with any stamp: %d; rows with stamp 0: %d; rows with some stamp but not stamp 0: %r" % (anyst.sum(), used.sum(), np.where(anyst & ~used)[0].tolist()))
for i in np.where(anyst & ~used)[0]:
    print("   row %d: %r" % (i, {c: int(v) for c, v in enumerate(st[i]) if v}))
print("rows present: first %d, last %d, missing inside: %r" % (np.where(anyst)[0].min(), np.where(anyst)[0].max(), sorted(set(range(int(np.where(anyst)[0].max()) + 1)) - set(np.where(anyst)[0].tolist()))))
st = st[used]
t0 = st[:, 0].min()
info = st[:, 28].copy()
found = st[:, 23].copy()
st[:, 28] = 0
st[:, 23] = 0
print("rounds entered (next round number) histogram: %r; workgroups with a full wave store: %d; still unresolved after the LDS rounds: %d" % (
    np.unique(info >> 32, return_counts=True), int(((info >> 16) & 1).sum()), int((info & 1).sum())))
print("keys kept by wave 0 of each workgroup: median %d, max %d" % (np.median(found), found.max()))
us = lambda a: (a - t0) / 100.0  # noqa: E731
print("workgroups with stamps: %d" % len(st))
cols = [(0, "start"), (27, "w15:start"), (11, "smp_req"), (19, "w15:smp_req"), (12, "sample"), (13, "slabs_req"), (8, "plan8"), (9, "plan9"), (10, "plan10"), (2, "plan"), (15, "sw15"), (3, "swept"), (4, "arrived"), (5, "last:begin"), (6, "last:adv"), (20, "alone1"), (21, "alone2"),
        (22, "alone3"), (24, "r2:flushed"), (25, "r3:flushed"), (26, "r4:flushed"), (7, "end")]
# (the second launch's stamps 4 / 5 / 6 overwrite the first's: with candidates, "arrived" .. "last:adv" are the second launch's)
for c, name in cols:
    v = st[:, c]
    ok = v > 0
    if ok.any():
        u = us(v[ok])
        print("%-11s n=%4d  min %6.1f  median %6.1f  p90 %6.1f  max %6.1f us" % (name, ok.sum(), u.min(), np.median(u), np.percentile(u, 90), u.max()))
# the slowest last arrivers
last = np.where(st[:, 5] > 0)[0]
order = np.argsort(-st[:, 7])[:6]
for i in order:
    print("wg row %4d: " % i + "  ".join("%s %.1f" % (n, us(st[i, c])) for c, n in cols if st[i, c] > 0))
