"""Dev tool (round 4): per-workgroup timeline of gptq_strip_kernel at the LLaMA-7B decode shapes, B = 1, 4-bit g128,
HBM-cold (rotating weight copies).
  python tools/lab/gptq_stamps.py build   (here: library with -DSBQ_GPTQ_STAMPS=1 -> tools/lab/libsbq_gptq_stamps.so)
  python tools/lab/gptq_stamps.py         (on the GPU box)
stamps (thread 0 of each workgroup, s_memrealtime 100 MHz): 0 start, 1 loads of the (first) pass issued, 2 first pass
landed (prefetching kernel) / barrier before the x staging passed (others), 3 last later pass landed (prefetching) /
x staged = x landed (others), 4 arithmetic done, 5 K-lane barrier passed, 6 result stored, 7 split fold done.
KNOB2=24: the 512-thread single-pass variant."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libsbq_gptq_stamps.so")

if len(sys.argv) > 1 and sys.argv[1] == "build":
    from sparsebit_amd import build as B
    B.build()
    obj = "/tmp/gptq_stamps.o"
    subprocess.check_call([B._hipcc()] + B.FLAGS + ["-DSBQ_GPTQ_STAMPS=1", "-c", os.path.join(B.CSRC, "sbq_gptq.hip"), "-o", obj])
    regular = [os.path.join(B.OBJ, f[:-4] + ".o") for f in B.sources() if f != "sbq_gptq.hip"]
    for src, units in B.EXTRA_UNITS.items():
        regular += [os.path.join(B.OBJ, src[:-4] + suffix + ".o") for suffix, _ in units]
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + regular + [obj])
    print("built", so)
    sys.exit(0)

import ctypes
import numpy as np
import torch
from sparsebit_amd import lib as L

L.LIB_PATH = so
lib = L.load()
dev = torch.device("cuda:0")
st = L.stream_ptr(dev)
stream = torch.cuda.current_stream(dev)
raw = ctypes.CDLL(so)
raw.sbq_debug_gptq_stamps.argtypes = [ctypes.c_void_p]
g = torch.Generator().manual_seed(1)
names = ["start", "loads issued", "pass 0 landed | barrier", "last pass landed | x staged", "arithmetic done", "K-lane barrier",
         "result stored", "split fold done"]
BATCH = int(os.environ.get("BATCH", "1"))
if BATCH >= 5:  # gptq_mfma_kernel's stamps (round 5)
    names = ["start", "block 0 + 1 loads issued", "block 0 computed", "all blocks computed", "partial tile published",
             "arrival counted", "fold done (last arriver)", "-"]
K2 = int(os.environ.get("KNOB2", "0"))
K1 = int(os.environ.get("KNOB1", "0"))
L.set_tuning(2, K2)
L.set_tuning(1, K1)
print("knob 2 = %d, knob 1 = %d, batch = %d" % (K2, K1, BATCH))
for in_f, out_f in ((4096, 4096), (4096, 11008), (11008, 4096)):
    groups = in_f // 128
    wb = in_f // 8 * out_f * 4
    copies = max(2, int(3.2e8 // wb) + 1)
    qws = [torch.randint(-2**31, 2**31 - 1, (in_f // 8, out_f), generator=g, dtype=torch.int64).to(torch.int32).to(dev) for _ in range(copies)]
    sc = (torch.rand(out_f, groups, generator=g) * 0.02 + 0.001).to(dev)
    zr = (torch.rand(out_f, groups, generator=g) * 0.1).to(dev)
    x = torch.randn(BATCH, in_f, generator=g).to(dev)
    y = torch.zeros(BATCH, out_f, device=dev)
    ws = L.fresh_workspace(lib.sbq_gptq_workspace_bytes(BATCH, in_f, out_f), dev)
    stamps = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)

    def run(i):
        lib.sbq_vecquant4matmul(L.ptr(x), L.ptr(qws[i % copies]), L.ptr(y), L.ptr(sc), L.ptr(zr), BATCH, in_f, out_f, 128, L.ptr(ws), ws.numel(), st)

    raw.sbq_debug_gptq_stamps(None)
    for i in range(50): run(i)
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for i in range(300): run(i)
    b.record(stream); torch.cuda.synchronize()
    avg = a.elapsed_time(b) * 1e3 / 300
    raw.sbq_debug_gptq_stamps(ctypes.c_void_p(stamps.data_ptr()))
    for i in range(7): run(i)
    torch.cuda.synchronize()
    raw.sbq_debug_gptq_stamps(None)
    s = stamps.reshape(-1, 8).cpu().numpy()
    s = s[s[:, 0] > 0]
    t0 = s[:, 0].min()
    print("== %d -> %d, %d workgroups; back-to-back launch interval (events, stamps off) %.2f us; us after the first "
          "workgroup's start: min / median / max" % (in_f, out_f, len(s), avg))
    for j in range(8):
        col = s[:, j]
        col = col[col >= t0]
        if len(col) == 0:
            continue
        rel = (col - t0) * 0.01
        print("  %-28s %5.2f / %5.2f / %5.2f   (%d workgroups)" % (names[j], rel.min(), np.median(rel), rel.max(), len(col)))
    del qws
