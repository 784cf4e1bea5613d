"""Dev tool: per-workgroup timeline of h16_select_kernel (library built by tools/lab/build_stamps.py; knob 1 == 779).
stamps: 0 start, 12 loads issued, 13 LDS cleared, 1 plan done (wave 0), 2 histogram complete (barrier), 3 windows binned +
flushed, 4 arrival known, 5 (last arriver) counter reset, 16/17 its gather / placement, 6 advance done, 7 end"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from sparsebit_amd import lib as L  # noqa: E402

L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsbq_stamps.so")
lib = L.load()
dev = torch.device("cuda:0")
st = L.stream_ptr(dev)
R = C = 4096
g = torch.Generator().manual_seed(0)
w = (torch.randn(R, C, generator=g) * torch.logspace(-2, 1, R).unsqueeze(1))
if os.environ.get("SEL_DATA") == "relu":
    w = torch.relu(torch.randn(R, C, generator=g))
w = w.bfloat16().to(dev)
xs = [w] + [torch.roll(w, i, 1).contiguous() for i in range(1, 12)]
n = R * C
sw = L.fresh_workspace(lib.sbq_radix_select_workspace_bytes(1, 2), dev)
out = torch.zeros(2, dtype=torch.float32, device=dev)
OLD = 256 + 64 * 128 + 8 * 2 * 2048 * 4 + 256
ONE = 256 + 64 * 128 + 8 * 2 * 2048 * 4


def run(kind, i):
    x = xs[i % 12]
    if kind == "kth":
        lib.sbq_kth_value(L.ptr(x), L.BF16, n, 1, n // 2 + 1, L.ptr(out), L.ptr(sw), sw.numel(), st)
    else:
        p = (ctypes.c_void_p * 1)(x.data_ptr())
        o = (ctypes.c_int64 * 1)(1)
        lib.sbq_percentile_select(p, o, 1, L.BF16, 1, n, 1e-3, L.ptr(out[0:1]), L.ptr(out[1:2]), L.ptr(sw), sw.numel(), st)


names = {0: "start", 12: "loads issued", 13: "LDS cleared", 1: "plan done (wave 0)", 15: "wave 1 counted", 18: "wave 15 counted", 11: "wave 0 counted",
         2: "histogram complete", 8: "windows binned", 9: "counters reduced", 3: "binned + flushed",
         4: "arrival known", 5: "counter reset", 16: "adv gathered", 17: "adv placed", 6: "advance done", 7: "end"}
for kind in ("kth", "pct"):
    for i in range(30):
        run(kind, i)
    torch.cuda.synchronize()
    L.set_tuning(1, 779)
    for i in range(5):
        run(kind, i)
    torch.cuda.synchronize()
    L.set_tuning(1, 0)
    s = sw[OLD + ONE: OLD + ONE + 256 * 256].view(torch.int64).reshape(256, 32).cpu().numpy().astype(np.int64)
    t0 = s[:, 0].min()
    rel = (s - t0) * 0.01
    print("== %s: us after the first workgroup's start: min / median / max over 256 workgroups" % kind)
    for j in (0, 12, 13, 1, 15, 18, 11, 2, 8, 9, 3, 4, 5, 16, 17, 6, 7):
        col = rel[:, j][s[:, j] >= t0]
        if j in (5, 6, 16, 17):
            last = np.argmax(s[:, 6])
            print("  %-22s last arriver (wg %d): %.2f" % (names[j], last, rel[last, j]))
        elif col.size:
            print("  %-22s %.2f / %.2f / %.2f" % (names[j], col.min(), np.median(col), col.max()))
