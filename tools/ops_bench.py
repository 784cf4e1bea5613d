"""Dev tool: time every op of the hot path at the headline sizes (GPU box)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sparsebit_amd import ops, lib as L
from sparsebit_amd.config import quantizer_config, sparser_config
from sparsebit_amd.observers import build_observer
from sparsebit_amd.quantizers.quant_descriptor import QuantDescriptor
from sparsebit_amd.sparsers import build_sparser
from sparsebit_amd import gptq

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
R = C = 4096
w = (torch.randn(R, C, generator=g) * torch.logspace(-2, 1, R).unsqueeze(1)).bfloat16()
NB = 10
xs = [torch.roll(w, i, 1).contiguous().to(dev) for i in range(NB)]
n = R * C

def timed(fn, iters=50, warm=5):
    for i in range(warm): fn(i)
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters

res = {}
def rep(name, us, bytes_):
    res[name] = {"us": round(us, 2), "GBps": round(bytes_ / us / 1e3, 1)}
    print("%-34s %9.2f us  %8.1f GB/s (algorithmic)" % (name, us, bytes_ / us / 1e3), flush=True)

mn, mx, _ = ops.channel_stats(xs[0], 0, True)
scale, zp = ops.qparams_from_minmax(mn, mx, -128, 127, True)
rep("qdq bf16->bf16 per-channel", timed(lambda i: ops.fake_quant(xs[i % NB], scale, zp, -128, 127, 0, out_dtype=torch.bfloat16)), n * 4)
rep("qdq bf16->bf16 per-tensor", timed(lambda i: ops.fake_quant(xs[i % NB], scale[:1], zp[:1], -128, 127, 0, out_dtype=torch.bfloat16)), n * 4)
rep("qdq bf16->fp32 per-channel", timed(lambda i: ops.fake_quant(xs[i % NB], scale, zp, -128, 127, 0)), n * 6)
xf = [x.float() for x in xs[:4]]
rep("qdq fp32->fp32 per-channel", timed(lambda i: ops.fake_quant(xf[i % 4], scale, zp, -128, 127, 0)), n * 8)
s4q, z4q = ops.qparams_from_minmax(mn, mx, -8, 7, True)
rep("qdq bf16->bf16 + int8 levels", timed(lambda i: ops.fake_quant(xs[i % NB], scale, zp, -128, 127, 0, out_dtype=torch.bfloat16, return_q=torch.int8)), n * 5)
rep("qdq bf16->bf16 + packed int4", timed(lambda i: ops.fake_quant(xs[i % NB], s4q, z4q, -8, 7, 0, out_dtype=torch.bfloat16, return_q="int4")), n * 4.5)
rep("quantize only bf16->int8", timed(lambda i: ops.quantize_only(xs[i % NB], scale, zp, -128, 127, 0, torch.int8)), n * 3)
rep("quantize only bf16->packed int4", timed(lambda i: ops.quantize_only(xs[i % NB], s4q, z4q, -8, 7, 0, "int4")), n * 2.5)
q8s = [ops.quantize_only(x, scale, zp, -128, 127, 0, torch.int8) for x in xs]
q4s = [ops.quantize_only(x, s4q, z4q, -8, 7, 0, "int4") for x in xs]
rep("dequantize int8->bf16", timed(lambda i: ops.dequantize_linear(q8s[i % NB], scale, zp, out_dtype=torch.bfloat16)), n * 3)
rep("dequantize packed int4->bf16", timed(lambda i: ops.dequantize_linear(q4s[i % NB], s4q, z4q, shape=(R, C), packed_int4=True, out_dtype=torch.bfloat16)), n * 2.5)
rep("dequantize int8->fp32", timed(lambda i: ops.dequantize_linear(q8s[i % NB], scale, zp)), n * 5)
# the same through the C ABI directly (ops.* costs ~15 us of Python per call: host-bound above)
lib = L.load(); st = L.stream_ptr(dev)
xp = [L.ptr(x) for x in xs]; sp_, zp_, s4p, z4p = L.ptr(scale), L.ptr(zp), L.ptr(s4q), L.ptr(z4q)
q8p = [L.ptr(t) for t in q8s]; q4p = [L.ptr(t) for t in q4s]
y16 = [torch.empty(R, C, dtype=torch.bfloat16, device=dev) for _ in range(NB)]; y16p = [L.ptr(t) for t in y16]
fw = lib.sbq_quant_perchannel_forward; dqf = lib.sbq_dequantize_linear
rep("[C ABI] qdq bf16->bf16", timed(lambda i: fw(xp[i % NB], L.BF16, y16p[i % NB], L.BF16, None, L.Q_NONE, sp_, zp_, 1, R, C, -128, 127, 0, st), 300, 50), n * 4)
rep("[C ABI] qdq bf16->bf16 + int8", timed(lambda i: fw(xp[i % NB], L.BF16, y16p[i % NB], L.BF16, q8p[i % NB], L.Q_I8, sp_, zp_, 1, R, C, -128, 127, 0, st), 300, 50), n * 5)
rep("[C ABI] qdq bf16->bf16 + int4", timed(lambda i: fw(xp[i % NB], L.BF16, y16p[i % NB], L.BF16, q4p[i % NB], L.Q_I4, s4p, z4p, 1, R, C, -8, 7, 0, st), 300, 50), n * 4.5)
rep("[C ABI] quantize only bf16->int8", timed(lambda i: fw(xp[i % NB], L.BF16, None, L.BF16, q8p[i % NB], L.Q_I8, sp_, zp_, 1, R, C, -128, 127, 0, st), 300, 50), n * 3)
rep("[C ABI] quantize only bf16->int4", timed(lambda i: fw(xp[i % NB], L.BF16, None, L.BF16, q4p[i % NB], L.Q_I4, s4p, z4p, 1, R, C, -8, 7, 0, st), 300, 50), n * 2.5)
rep("[C ABI] dequantize int8->bf16", timed(lambda i: dqf(q8p[i % NB], L.Q_I8, 1, y16p[i % NB], L.BF16, sp_, zp_, 1, R, C, st), 300, 50), n * 3)
rep("[C ABI] dequantize int4->bf16", timed(lambda i: dqf(q4p[i % NB], L.Q_I4, 1, y16p[i % NB], L.BF16, s4p, z4p, 1, R, C, st), 300, 50), n * 2.5)
del q8s, q4s, y16
rep("minmax stats per-channel", timed(lambda i: ops.channel_stats(xs[i % NB], 0, True)), n * 2)
rep("minmax stats per-tensor", timed(lambda i: ops.channel_stats(xs[i % NB], 0, False)), n * 2)

def mse(i, perch):
    cfg = quantizer_config("per-%s-symmetric" % ("channel" if perch else "tensor"), 8, observer="MSE")
    o = build_observer(cfg, QuantDescriptor(cfg)); o.data_cache.update(xs[i % NB]); o.calc_qparams()
rep("MSE observer per-channel (e2e)", timed(lambda i: mse(i, True), 10, 2), n * 2)
rep("MSE observer per-tensor (e2e)", timed(lambda i: mse(i, False), 10, 2), n * 2)
sse = torch.zeros(C, 80, dtype=torch.float64, device=dev)
rep("  mse_accumulate kernel only", timed(lambda i: ops.mse_accumulate(xs[i % NB], mn, mx, -128, 127, True, sse), 10, 2), n * 2)

def pct(i, perch):
    cfg = quantizer_config("per-%s-symmetric" % ("channel" if perch else "tensor"), 8, observer="PERCENTILE")
    o = build_observer(cfg, QuantDescriptor(cfg)); o.data_cache.update(xs[i % NB]); o.calc_minmax()
rep("percentile per-channel rows (e2e)", timed(lambda i: pct(i, True), 20, 2), n * 2)
rep("percentile per-tensor radix (e2e)", timed(lambda i: pct(i, False), 10, 2), n * 2 * 4)
sp = build_sparser(sparser_config(0.5))
rep("l1 threshold (3-pass radix)", timed(lambda i: sp.calc_threshold(xs[i % NB]), 10, 2), n * 2 * 3)
thr = sp.calc_threshold(xs[0])
rep("mask_from_threshold", timed(lambda i: ops.mask_from_threshold(xs[i % NB], thr)), n * 3)
mask = ops.mask_from_threshold(xs[0], thr)
s4 = (2 * xs[0].float().abs().mean(1) / 7 ** 0.5).contiguous(); z4 = torch.zeros_like(s4)
rep("fused mask(bytes)+QDQ bf16->bf16", timed(lambda i: ops.fake_quant(xs[i % NB], s4, z4, -8, 7, 0, out_dtype=torch.bfloat16, mask=mask)), n * 5)
rep("fused thresh+QDQ bf16->bf16", timed(lambda i: ops.fake_quant(xs[i % NB], s4, z4, -8, 7, 0, out_dtype=torch.bfloat16, thresh=thr)), n * 4)
gy = [torch.randn(R, C, device=dev).bfloat16() for _ in range(2)]
rep("STE backward bf16 (gx+gs+gzp)", timed(lambda i: ops.fake_quant_backward(xs[i % NB], gy[i % 2], s4, z4, -8, 7, 0)), n * 6)
rep("STE backward bf16 (gx only)", timed(lambda i: ops.fake_quant_backward(xs[i % NB], gy[i % 2], s4, z4, -8, 7, 0, False, False)), n * 6)

# GPTQ 4-bit g128 4096x4096
torch.manual_seed(0)
lin = torch.nn.Linear(4096, 4096).to(dev)
qz = gptq.Quantizer(); qz.configure(bit=4, perchannel=True, sym=False, mse=False)
qz.find_params(lin.weight.data, weight=True, groupsize=128)
lin.weight.data = gptq.quantize(lin.weight.data.view(-1, 128), qz.scale.view(-1, 1), qz.zero.view(-1, 1), qz.maxq).view(4096, 4096)
ql = gptq.QuantLinear(4096, 4096, 4, 128).to(dev); ql.pack(lin, qz.scale, qz.zero)
for B in (1, 8, 32):
    xb = torch.randn(B, 4096, device=dev)
    yb = ql.bias.expand(B, 4096).clone()
    sc, zr = ql.scales.float().reshape(-1).contiguous(), ql.zeros.float().reshape(-1).contiguous()
    bytes_ = 4096 * 512 * 4 + 2 * 4096 * 32 * 4 + B * (4096 * 4 + 2 * 4096 * 4)
    rep("gptq matvec B=%d (kernel)" % B, timed(lambda i: ops.vecquant4matmul(xb, ql.qweight, yb, sc, zr, 128), 100), bytes_)
    ref = torch.nn.functional.linear(xb, lin.weight, lin.bias)
    got = ql(xb)
    print("   max abs err vs dense fp32 linear: %.3g" % (got - ref).abs().max().item())
json.dump(res, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "ops_bench.json"), "w"), indent=1)
