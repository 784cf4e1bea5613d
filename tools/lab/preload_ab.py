"""Dev tool (round 4): kernel-argument preload A/B on ONE box -- the headline per-channel QDQ launch (4096 x 4096 bf16,
12 rotating tensors) through the product library (built with -amdgpu-kernarg-preload-count=16) and through a copy built
without it, alternating.
  python tools/lab/preload_ab.py build   (here)      python tools/lab/preload_ab.py   (GPU box)"""
import ctypes
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libsbq_nopreload.so")
from sparsebit_amd import build as B

if len(sys.argv) > 1 and sys.argv[1] == "build":
    B.build()
    flags = [f for f in B.FLAGS if f not in ("-mllvm", "-amdgpu-kernarg-preload-count=16")]
    objs = []
    for src in B.sources():
        units = [("", [])] + B.EXTRA_UNITS.get(src, [])
        for suffix, extra in units:
            obj = "/tmp/nopre_%s%s.o" % (src[:-4], suffix)
            subprocess.check_call([B._hipcc()] + flags + extra + ["-c", os.path.join(B.CSRC, src), "-o", obj])
            objs.append(obj)
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs)
    print("built", so)
    sys.exit(0)

import torch
from sparsebit_amd import lib as L

libA = L.load()
libB = ctypes.CDLL(so)
libB.sbq_quant_perchannel_forward.argtypes = libA.sbq_quant_perchannel_forward.argtypes
libB.sbq_quant_perchannel_forward.restype = ctypes.c_int
dev = torch.device("cuda:0")
st = L.stream_ptr(dev)
stream = torch.cuda.current_stream(dev)
R = C = 4096
g = torch.Generator().manual_seed(0)
w = (torch.randn(R, C, generator=g) * torch.logspace(-2, 1, R).unsqueeze(1)).bfloat16().to(dev)
xs = [w] + [torch.roll(w, i, 1).contiguous() for i in range(1, 12)]
ys = [torch.empty_like(x) for x in xs]
scale = (w.float().abs().amax(1) / 127).contiguous()
zp = torch.zeros(R, device=dev)


def run(lib, i):
    j = i % 12
    lib.sbq_quant_perchannel_forward(L.ptr(xs[j]), L.BF16, L.ptr(ys[j]), L.BF16, None, L.Q_NONE, L.ptr(scale), L.ptr(zp), 1, R, C,
                                     -128, 127, 0, st)


def window(lib, n=1024):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for i in range(n):
        run(lib, i)
    b.record(stream)
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


for lib in (libA, libB):
    for _ in range(3):
        window(lib)
res = {"preload": [], "no preload": []}
for rep in range(10):
    res["preload"].append(window(libA))
    res["no preload"].append(window(libB))
for k, v in res.items():
    v = sorted(v)
    print("%-11s us per launch over 10 windows of 1024: min %.3f median %.3f max %.3f" % (k, v[0], v[len(v) // 2], v[-1]))
