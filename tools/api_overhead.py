"""Dev tool: host-side cost of one quantizer call through the Python layers (tiny tensor, so the
kernel is negligible): ops.fake_quant, Quantizer.forward under no_grad, and with autograd."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sparsebit_amd import ops
from sparsebit_amd.common import Backend
from sparsebit_amd.config import quantizer_config
from sparsebit_amd.quantizers import build_quantizer

dev = "cuda"
w = torch.randn(64, 64, 3, 3, device=dev)
q = build_quantizer(quantizer_config("per-channel-symmetric", 8)); q.set_backend(Backend.VIRTUAL)
q.update_observer(w); q.calc_qparams(); q.enable_quant()
l = build_quantizer(quantizer_config("per-channel-symmetric", 4, quantizer="lsq")); l.set_backend(Backend.VIRTUAL)
l.update_observer(w); l.calc_qparams(); l.enable_quant()
s, z = q.scale, q.zero_point
wp = torch.nn.Parameter(w.clone())

def rate(fn, n=2000):
    for _ in range(200): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t = time.perf_counter() - t0; torch.cuda.synchronize()
    return t / n * 1e6

with torch.no_grad():
    print("ops.fake_quant            : %6.1f us/call" % rate(lambda: ops.fake_quant(w, s, z, -128, 127, 0)))
    print("uniform quantizer, no_grad: %6.1f us/call" % rate(lambda: q(w)))
    print("lsq quantizer, no_grad    : %6.1f us/call" % rate(lambda: l(w)))
print("uniform quantizer, autograd: %6.1f us/call (forward only)" % rate(lambda: q(wp)))
print("lsq quantizer, autograd    : %6.1f us/call (forward only)" % rate(lambda: l(wp)))
def fb():
    y = l(wp); y.sum().backward()
print("lsq forward+backward       : %6.1f us/call" % rate(fb, 500))
# a whole "model": 53 LSQ weight quantizers one by one vs grouped, forward + backward
from sparsebit_amd.group import WeightQuantGroup
shapes = []
inp = 64
for width, blocks in ((64, 3), (128, 4), (256, 6), (512, 3)):
    for b in range(blocks):
        shapes += [(width, inp, 1, 1), (width, width, 3, 3), (width * 4, width, 1, 1)]
        if b == 0:
            shapes.append((width * 4, inp, 1, 1))
        inp = width * 4
shapes.append((1000, 2048))
triples = []
for shp in shapes:
    p_ = torch.nn.Parameter(torch.randn(shp, device=dev) * 0.05)
    ql = build_quantizer(quantizer_config("per-channel-symmetric", 4, quantizer="lsq")); ql.set_backend(Backend.VIRTUAL)
    ql.update_observer(p_.detach()); ql.calc_qparams(); ql.enable_quant()
    triples.append((ql, p_, None))
group = WeightQuantGroup(triples)
gys = [torch.randn(shp, device=dev) for shp in shapes]
def step(fn):
    for q_, w_, _ in triples:  # optimizer.zero_grad(set_to_none=True)
        w_.grad = None
        q_.scale.grad = None
    outs = fn()
    torch.autograd.backward(outs, gys)
print("53 LSQ weight quantizers fwd+bwd, one by one: %8.1f us/step" % rate(lambda: step(lambda: [q_(w_) for q_, w_, _ in triples]), 30))
print("53 LSQ weight quantizers fwd+bwd, grouped   : %8.1f us/step" % rate(lambda: step(group), 30))
pr = cProfile.Profile(); pr.enable()
for _ in range(50): step(group)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
