"""Dev tool: per-workgroup timeline of the one-launch selection kernel (knob 1 == 779: s_memrealtime stamps, 100 MHz).
stamps: 0 start, 1 sample + slab loads issued / hist clear, 2 plan done, 3 sweep + flush done, 4 arrival known,
5 (last arriver) counter reset, 6 advance done, 7 end"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from sparsebit_amd import lib as L
L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsbq_stamps.so")  # built with -DSBQ_SEL_STAMPS=1
lib = L.load()
dev = torch.device("cuda:0")
st = L.stream_ptr(dev)
R = C = 4096
g = torch.Generator().manual_seed(0)
w = (torch.randn(R, C, generator=g) * torch.logspace(-2, 1, R).unsqueeze(1))
if os.environ.get("SEL_DATA") == "relu": w = torch.relu(torch.randn(R, C, generator=g))
w = w.bfloat16().to(dev)
xs = [w] + [torch.roll(w, i, 1).contiguous() for i in range(1, 12)]
n = R * C
nbytes = lib.sbq_radix_select_workspace_bytes(1, 2)
sw = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
out = torch.zeros(2, dtype=torch.float32, device=dev)
OLD = 256 + 64 * 128 + 8 * 2 * 2048 * 4 + 256
ONE = 256 + 64 * 128 + 8 * 2 * 2048 * 4
import ctypes
def run(kind, i):
    x = xs[i % 12]
    if kind == "kth":
        lib.sbq_kth_value(L.ptr(x), L.BF16, n, 1, n // 2 + 1, L.ptr(out), L.ptr(sw), sw.numel(), st)
    else:
        p = (ctypes.c_void_p * 1)(x.data_ptr()); o = (ctypes.c_int64 * 1)(1)
        lib.sbq_percentile_select(p, o, 1, L.BF16, 1, n, 1e-3, L.ptr(out[0:1]), L.ptr(out[1:2]), L.ptr(sw), sw.numel(), st)
for kind in ("kth", "pct"):
    for i in range(30): run(kind, i)
    torch.cuda.synchronize()
    L.set_tuning(1, 779)
    for i in range(5): run(kind, i)
    torch.cuda.synchronize()
    L.set_tuning(1, 0)
    s = sw[OLD + ONE: OLD + ONE + 256 * 256].view(torch.int64).reshape(256, 32).cpu().numpy().astype(np.int64)
    t0 = s[:, 0].min()
    rel = (s - t0) * 0.01  # us
    names = ["start", "loads issued", "plan done", "sweep+flush", "arrival", "reset", "advance", "end", "plan: hist", "plan: scan", "plan: ranks", "hist cleared", "sample issued", "slabs issued", "sample in", "sweep loop done", "adv0 gathered", "adv0 placed", "adv1 gathered", "adv1 placed", "counters reduced"]
    w0 = s[0, 30]; w1 = s[0, 31]; sh = s[0, 29]
    print("   windows of workgroup 0: sel0 lo=%08x span=%08x shift=%d | sel1 lo=%08x span=%08x shift=%d | all workgroups agree: %s" % (
        w0 & 0xffffffff, (w0 >> 32) & 0xffffffff, sh & 0xffffffff, w1 & 0xffffffff, (w1 >> 32) & 0xffffffff, (sh >> 32) & 0xffffffff,
        bool((s[:, 30] == w0).all() and (s[:, 31] == w1).all())))
    print("== %s: per-stamp us after the first workgroup's start: min / median / max over 256 workgroups" % kind)
    for j, nm in enumerate(names):
        col = rel[:, j]
        col = col[s[:, j] >= t0]  # stamps 5, 6 exist for the last arriver only (older values otherwise)
        if nm == '-' or (col.size == 0 and j not in (5, 6, 16, 17, 18, 19)): continue
        if j in (5, 6, 16, 17, 18, 19):
            last = np.argmax(s[:, 6])  # the workgroup that advanced: the only fresh stamp 6
            print("  %-14s last arriver (wg %d): %.2f" % (nm, last, rel[last, j]))
        else:
            print("  %-14s %.2f / %.2f / %.2f" % (nm, col.min(), np.median(col), col.max()))
