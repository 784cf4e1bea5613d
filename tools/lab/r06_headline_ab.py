"""lab: the headline launch (per-channel int8 QDQ of a 4096 x 4096 bf16 weight) and the parity-mode launch (fp32 out)
under the library named by SBQ_LIB -- HIP events over 3 x 3000 launches on rotating buffers."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

from sparsebit_amd import lib as L  # noqa: E402

if os.environ.get("SBQ_LIB"):
    L.LIB_PATH = os.path.abspath(os.environ["SBQ_LIB"])
lib = L.load()
dev = torch.device("cuda:0")
R = C = 4096
xs = [torch.randn(R, C, device=dev).bfloat16() for _ in range(12)]
ys = [torch.empty_like(x) for x in xs]
yf = [torch.empty(R, C, device=dev) for _ in range(6)]
scale = (torch.rand(R, device=dev) * 0.01 + 0.001).float()
zp = torch.zeros(R, device=dev)
st = L.stream_ptr(dev)


def run(i, f32):
    x = xs[i % 12]
    y = yf[i % 6] if f32 else ys[i % 12]
    rc = lib.sbq_quant_perchannel_forward(L.ptr(x), L.BF16, L.ptr(y), L.F32 if f32 else L.BF16, None, 0, L.ptr(scale), L.ptr(zp), 1, R, C, -128, 127, 0, st)
    assert rc == 0, rc


def timed(f32, n=3000):
    for i in range(300):
        run(i, f32)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        run(i, f32)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


sc_o, zp_o, mn_o, mx_o = (torch.empty(R, device=dev) for _ in range(4))
ws = L.fresh_workspace(1 << 20, dev)


def run_obs(i, f32):
    x = xs[i % 12]
    y = yf[i % 6] if f32 else ys[i % 12]
    rc = lib.sbq_observe_quant_perchannel_forward(L.ptr(x), L.BF16, L.ptr(y), L.F32 if f32 else L.BF16, L.ptr(sc_o), L.ptr(zp_o), L.ptr(mn_o),
                                                  L.ptr(mx_o), R, C, -128, 127, 1, L.ptr(ws), ws.numel(), st)
    assert rc == 0, rc


def timed_obs(f32, n=3000):
    for i in range(300):
        run_obs(i, f32)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        run_obs(i, f32)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


print(os.path.basename(L.LIB_PATH), "fused observe:", " ".join("bf16 out %.2f us, fp32 out %.2f us |" % (timed_obs(False), timed_obs(True)) for _ in range(2)), flush=True)
print(os.path.basename(L.LIB_PATH), " ".join("bf16 out %.2f us, fp32 out %.2f us |" % (timed(False), timed(True)) for _ in range(3)), flush=True)
