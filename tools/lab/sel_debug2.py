"""Dev tool: catch a selection that does not resolve (knob 1 == 778 keeps the state after the lonely rounds)."""
import os, sys, struct, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from sparsebit_amd import lib as L
lib = L.load()
dev = torch.device("cuda:0")
st = L.stream_ptr(dev)
OLD = 256 + 64 * 128 + 8 * 2 * 2048 * 4 + 256
KROT = 0x007fffff
def key_abs(xf):
    b = np.abs(xf).astype(np.float32).view(np.uint32)
    return ((b & 0x7fffffff) | 0x80000000) - KROT
def dump(sw, tag, keys, k):
    raw = bytes(sw[OLD:OLD + 96].cpu().numpy())
    lo, shift, span, side, kk, done, fresh = struct.unpack_from("<IIIIqII", raw, 0)
    nn, arrivals = struct.unpack_from("<qI", raw, 64)
    inside = int(((keys >= lo) & (keys.astype(np.int64) <= lo + span)).sum())
    below = int((keys < lo).sum())
    hist = sw[OLD + 256 + 64 * 128: OLD + 256 + 64 * 128 + 8 * 2 * 2048 * 4].view(torch.int32).reshape(8, 2, 2048).cpu()
    slots = sw[OLD + 256: OLD + 256 + 64 * 128].view(torch.int64).reshape(64, 16).cpu()
    print("%s: lo=%08x shift=%d span=%08x k=%d done=%d fresh=%d n=%d arrivals=%d | true: below=%d inside=%d -> k - below = %d | hist sum=%d below sum=%d"
          % (tag, lo, shift, span, kk, done, fresh, nn, arrivals, below, inside, k - below, int(hist.sum()), int(slots[:, 0].sum())), flush=True)

L.set_tuning(1, 778)
# (1) the failing test case
n = 3 * 16384 + 5
g = torch.Generator().manual_seed(n)
x = (torch.randn(n, generator=g) * torch.rand(n, generator=g) * 8).to(torch.bfloat16)
keys = key_abs(x.float().numpy())
xd = x.cuda()
for k in (1, n // 2, n, 1):
    sw = torch.zeros(lib.sbq_radix_select_workspace_bytes(1, 2), dtype=torch.uint8, device=dev)
    out = torch.full((2,), -1.0, dtype=torch.float32, device=dev)
    lib.sbq_kth_value(L.ptr(xd), L.BF16, n, 1, k, L.ptr(out), L.ptr(sw), sw.numel(), st)
    torch.cuda.synchronize()
    ref = np.sort(np.abs(x.float().numpy()))[k - 1]
    dump(sw, "n=%d k=%d out=%r ref=%r" % (n, k, out[0].item(), float(ref)), keys, k)
# (2) the big tensor, median, until a slow call shows up
R = C = 4096
g = torch.Generator().manual_seed(0)
w = (torch.randn(R, C, generator=g) * torch.logspace(-2, 1, R).unsqueeze(1)).bfloat16()
keys = key_abs(w.float().numpy().reshape(-1))
wd = w.to(dev)
n = R * C
k = n // 2 + 1
caught = 0
for it in range(60):
    sw = torch.zeros(lib.sbq_radix_select_workspace_bytes(1, 2), dtype=torch.uint8, device=dev)
    out = torch.full((2,), -1.0, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lib.sbq_kth_value(L.ptr(wd), L.BF16, n, 1, k, L.ptr(out), L.ptr(sw), sw.numel(), st)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) * 1e6
    if us > 500 or it < 2:
        dump(sw, "big it=%d %.0f us out=%r" % (it, us, out[0].item()), keys, k)
        caught += us > 500
        if caught >= 3: break
print("slow calls caught:", caught)
L.set_tuning(1, 0)
