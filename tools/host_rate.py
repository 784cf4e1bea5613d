"""Dev tool: is the single-launch loop host-bound?  Measures host enqueue time per launch."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sparsebit_amd import lib as L
lib = L.load()
dev = torch.device("cuda:0")
R = C = 4096
xs = [torch.randn(R, C, device=dev).bfloat16() for _ in range(12)]
ys = [torch.empty_like(x) for x in xs]
scale = (xs[0].float().abs().amax(1) * 2 / 255).contiguous(); zp = torch.zeros_like(scale)
st = L.stream_ptr(dev)
xp = [L.ptr(x) for x in xs]; yp = [L.ptr(y) for y in ys]; sp, zpp = L.ptr(scale), L.ptr(zp)
fwd = lib.sbq_quant_perchannel_forward
def step(i):
    j = i % 12
    fwd(xp[j], 2, yp[j], 2, None, 0, sp, zpp, 1, R, C, -128, 127, 0, st)
for rep in range(4):
    for i in range(200): step(i)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for i in range(2000): step(i)
    t_host = time.perf_counter() - t0
    e1.record(); torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("rep %d: host enqueue %.2f us/launch, wall %.2f us/launch, events %.2f us/launch" % (rep, t_host / 2000 * 1e6, t_all / 2000 * 1e6, e0.elapsed_time(e1) / 2000 * 1e3), flush=True)
# tiny kernel launch rate for reference
one = torch.zeros(8, device=dev).bfloat16(); o2 = torch.empty_like(one); s1 = torch.ones(1, device=dev); z1 = torch.zeros(1, device=dev)
f2 = lib.sbq_quant_pertensor_forward
t0 = time.perf_counter()
for i in range(2000): f2(L.ptr(one), 2, L.ptr(o2), 2, None, 0, L.ptr(s1), L.ptr(z1), 8, -128, 127, 0, st)
t_host = time.perf_counter() - t0; torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print("tiny kernel: host %.2f us/launch, wall %.2f us/launch" % (t_host / 2000 * 1e6, t_all / 2000 * 1e6))
print("cpu count", os.cpu_count())
