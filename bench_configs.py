"""The BASELINE.json configs beside the headline, as timed legs with in-run oracle gates.

bench.py times the headline (per-channel int8 QDQ of a 4096 x 4096 bf16 weight) in its timed region; everything
here runs AFTER that region, on rank 0's GPU as well as every other rank's, and lands in the JSON line's
`extras["configs"]`.  Each leg reports

    us                  average launch-to-launch time of the library call(s), HIP events on the launch stream
    algorithmic_bytes   SURVEY.md 8(d)'s per-unit figure x the units one call processes
    GBps, frac          algorithmic_bytes / us, and that over the 8 TB/s HBM3E peak
    parity, gate        a boolean oracle check computed in this very run, and what it compared

so that BENCH_rNN.json alone answers "how fast, against which roof, and is it right" for configs 1-5.  The oracle
(oracle/: the reference's CPU algorithm, pinned to the reference's own outputs by tests/golden/) is used here as
the CHECKER only; every timed call goes through the C ABI of libsbq.so with buffers that are already in HBM, and
inputs rotate through more than the 256 MiB Infinity Cache wherever the working set is smaller than that.

Reference ops behind the legs:
  config 1  observers/minmax.py:14-25 (streaming, per tensor), quantizers/quant_tensor.py:128-156 (TensorRT backend)
  config 2  observers/mse.py:28-63 (80 candidates per channel)
  config 3  observers/percentile.py:16-46 over the cached calibration batches (DeiT-small: 64 x 197 x 384)
  config 4  large_language_models/llama/quantization/utils/quant.py:281-307 -> cuda/cuda_kernel_4bit.cu:36-180
  config 5  sparse/sparsers/l1norm.py:18-26, sparse/modules/conv.py:39-43 + quantizers/lsq.py:24-76,
            torch_extensions/fake_quant_tensor.cu:227-270 (backward)
"""
import ctypes
import math

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0
FP32_VECTOR_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 (vector)
ROWS = COLS = 4096


def _entry(us, nbytes, parity, gate, **kw):
    d = {
        "us": round(us, 3),
        "algorithmic_bytes": int(nbytes),
        "GBps": round(nbytes / us / 1e3, 1),
        "frac": round(nbytes / us / 1e3 / HBM_PEAK_GBS, 4),
        "parity": None if parity is None else bool(parity),
        "gate": gate,
    }
    d.update(kw)
    return d


class Ctx:
    """what bench.py hands over: device, library, the 12 rotating 4096 x 4096 bf16 weights, the host copy of weight 0"""

    def __init__(self, dev, lib, L, ops, stream, host, xs, ys, scale, zp):
        self.dev, self.lib, self.L, self.ops, self.stream = dev, lib, L, ops, stream
        self.st = L.stream_ptr(dev)
        self.host, self.xs, self.ys, self.scale, self.zp = host, xs, ys, scale, zp
        self.n = ROWS * COLS

    def timed(self, fn, iters, warm=10, rounds=2):
        """average us per call of fn(i): events on the launch stream, best of `rounds` loops"""
        best = float("inf")
        for _ in range(rounds):
            for i in range(warm):
                fn(i)
            torch.cuda.synchronize(self.dev)
            a = torch.cuda.Event(enable_timing=True)
            b = torch.cuda.Event(enable_timing=True)
            a.record(self.stream)
            for i in range(iters):
                fn(i)
            b.record(self.stream)
            torch.cuda.synchronize(self.dev)
            best = min(best, a.elapsed_time(b) * 1e3 / iters)
        return best


class _With:
    """a Ctx with some attributes replaced (a leg that brings its own tensors)"""

    def __init__(self, base, **kw):
        self.__dict__.update(base.__dict__)
        self.__dict__.update(kw)
        self.timed = base.timed


def _same(a, b):
    """bit-for-bit equal, NaN == NaN, +0 == -0"""
    a = np.asarray(a)
    b = np.asarray(b)
    return bool(a.shape == b.shape and np.all((a == b) | (np.isnan(a) & np.isnan(b))))


def _bf16_np(t):
    """torch bf16 tensor -> numpy fp32 with the same values"""
    return t.float().cpu().numpy()


def _as_bf16_bits(a):
    """numpy fp32 -> the bf16 value RNE(a) as fp32 (what a bf16 output must equal bit for bit)"""
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).bfloat16().float().numpy()


GATE_ROWS = list(range(0, ROWS, 64)) + [1, 255, 4095]  # 67 rows of the 4096


# ------------------------------------------------------------------------------------------------------
# headline gates: SURVEY.md 8(d) "parity gates run in the same benchmark"
# ------------------------------------------------------------------------------------------------------
def headline_gates(c):
    from oracle import oracle as O

    L, lib, ops = c.L, c.lib, c.ops
    x = c.xs[0]
    rows = GATE_ROWS
    xf = c.host.float().numpy()
    # observer (minmax.py:14-25) and qparams (base.py:63-79): bit-exact on ALL rows
    mn, mx, _ = ops.channel_stats(x, 0, True)
    mn_ref, mx_ref = O.minmax(xf, 0, True)
    s_ref, z_ref = O.qparams_from_minmax(mn_ref, mx_ref, -128, 127, True)
    g_minmax = bool(np.array_equal(mn.cpu().numpy(), mn_ref) and np.array_equal(mx.cpu().numpy(), mx_ref))
    g_qparams = bool(np.array_equal(c.scale.cpu().numpy(), s_ref) and np.array_equal(c.zp.cpu().numpy(), z_ref))
    # integer levels + fp32 dequantized output of the timed entry point (parity mode: bf16 in, fp32 + int8 out)
    y32 = torch.empty(ROWS, COLS, dtype=torch.float32, device=c.dev)
    q8 = torch.empty(ROWS, COLS, dtype=torch.int8, device=c.dev)
    rc = lib.sbq_quant_perchannel_forward(L.ptr(x), L.BF16, L.ptr(y32), L.F32, L.ptr(q8), L.Q_I8, L.ptr(c.scale), L.ptr(c.zp),
                                          1, ROWS, COLS, -128, 127, 0, c.st)
    L.check(rc)
    rc = lib.sbq_quant_perchannel_forward(L.ptr(x), L.BF16, L.ptr(c.ys[0]), L.BF16, None, L.Q_NONE, L.ptr(c.scale), L.ptr(c.zp),
                                          1, ROWS, COLS, -128, 127, 0, c.st)
    L.check(rc)
    torch.cuda.synchronize(c.dev)
    dq_ref, q_ref = O.qdq(xf[rows], s_ref[rows], z_ref[rows], -128, 127, 0)
    g_q = bool(np.array_equal(q8[rows].cpu().numpy().astype(np.int32), q_ref))
    got = y32[rows].cpu().numpy()
    den = np.where(dq_ref == 0, 1.0, np.abs(dq_ref))
    max_rel = float(np.max(np.abs(got - dq_ref) / den))
    g_dq = bool(max_rel <= 1e-6)
    g_bf16 = bool(np.array_equal(c.ys[0][rows].float().cpu().numpy(), _as_bf16_bits(dq_ref)))
    return {
        "rows_checked": len(rows),
        "q_int_bit_exact": g_q,
        "dq_f32_max_rel_err": max_rel,
        "dq_f32_within_1e-6": g_dq,
        "dq_f32_bit_exact": bool(np.array_equal(got, dq_ref)),
        "bf16_out_equals_rne_of_ref": g_bf16,
        "observer_minmax_bit_exact_all_rows": g_minmax,
        "scale_zp_bit_exact_all_rows": g_qparams,
        "all": bool(g_q and g_dq and g_bf16 and g_minmax and g_qparams),
    }


# ------------------------------------------------------------------------------------------------------
# config 1: ResNet-18 PTQ 8w8a, min-max observers, TensorRT backend (examples/post_training_quantization/imagenet1k/
# basecase/qconfig.yaml: W per-channel-symmetric, A per-tensor-symmetric, fp32 model) -- the GPU side of the
# reference's own CPU-runnable case: streaming per-tensor min-max over the calibration batches of the largest
# activation, per-tensor int8 QDQ of that activation at batch 256, per-channel QDQ of the largest conv weight
# ------------------------------------------------------------------------------------------------------
def config1_resnet18_minmax(c):
    from oracle import oracle as O

    L, lib, ops = c.L, c.lib, c.ops
    out = {}
    g = torch.Generator().manual_seed(11)
    # -- (a) the observer: four calibration batches 64 x 64 x 56 x 56 fp32 of the stem's output (observers/minmax.py:14-25
    #    per tensor; calibration.py:109-115 feeds them batch by batch).  Eight batches rotate (411 MB > Infinity Cache).
    shape = (64, 64, 56, 56)
    n_b = shape[0] * shape[1] * shape[2] * shape[3]
    host_b = [torch.relu(torch.randn(shape, generator=g)) * (1.0 + 0.25 * i) for i in range(2)]
    batches = []
    for i in range(8):
        b = host_b[i % 2].to(c.dev)
        batches.append(b if i < 2 else torch.roll(b.reshape(-1), i).reshape(shape).contiguous())
    mn_o = torch.empty(8, dtype=torch.float32, device=c.dev)
    mx_o = torch.empty(8, dtype=torch.float32, device=c.dev)
    ws = torch.empty(max(lib.sbq_stats_workspace_bytes(1, 1, n_b), 16), dtype=torch.uint8, device=c.dev)
    states = [ops.minmax_state(c.dev) for _ in range(8)]

    def stream4(i):
        # the product path (observers/minmax.py consume): ONE launch per batch, the running state updated in place
        for j in range(4):
            k = (4 * i + j) % 8
            lib.sbq_minmax_accumulate(L.ptr(batches[k]), L.F32, n_b, L.ptr(states[k]), c.st)

    def stats4(i):
        # round 3's route: chunk partials + fold launch per batch (and a torch.minimum / maximum pair on top of it)
        for j in range(4):
            k = (4 * i + j) % 8
            lib.sbq_channel_stats(L.ptr(batches[k]), L.F32, 1, 1, n_b, L.ptr(mn_o[k:k + 1]), L.ptr(mx_o[k:k + 1]), None, L.ptr(ws),
                                  ws.numel(), c.st)

    us = c.timed(stream4, 20, warm=4)
    us_two_launch = c.timed(stats4, 20, warm=4)
    stats4(0)
    stats4(1)
    torch.cuda.synchronize(c.dev)
    ok = True
    for k in range(8):
        ref = host_b[k % 2].numpy()
        lo, hi = ops.minmax_state_read(states[k])  # every batch went into its own state: per-batch extrema
        ok = ok and float(lo) == float(ref.min()) and float(hi) == float(ref.max())
        ok = ok and float(mn_o[k]) == float(ref.min()) and float(mx_o[k]) == float(ref.max())
    # the streaming fold itself, through the product observer (consume: statistics kernel + minimum / maximum)
    from sparsebit_amd.common import Backend
    from sparsebit_amd.config import quantizer_config
    from sparsebit_amd.quantizers import build_quantizer

    qa = build_quantizer(quantizer_config("per-tensor-symmetric", 8, observer="MINMAX", target="feature", layout="NCHW"))
    qa.set_backend(Backend.TENSORRT)
    qa.dims = 4
    for k in range(4):
        qa.observer.consume(batches[k])
    s_a, z_a = qa.calc_qparams()
    mn_r = min(float(host_b[k % 2].min()) for k in range(4))
    mx_r = max(float(host_b[k % 2].max()) for k in range(4))
    s_ref, z_ref = O.qparams_from_minmax(np.array([mn_r], np.float32), np.array([mx_r], np.float32), -128, 127, True)
    ok = ok and float(s_a) == float(s_ref[0]) and float(z_a) == float(z_ref[0])
    out["minmax_observer_4_batches_64x64x56x56_fp32"] = _entry(
        us, 4 * n_b * 4, ok, "min / max of every batch == numpy; streaming observer's (scale, zp) over 4 batches == oracle "
        "(observers/minmax.py:14-25 + base.py:63-79)", elements=4 * n_b, launches=4,
        sbq_channel_stats_two_launches_per_batch_us=round(us_two_launch, 2))
    # -- (b) the quantizer on the activation at the reference's calibration batch size 256 (256 x 64 x 56 x 56 =
    #    51.4 M fp32 elements): per-tensor symmetric int8, TensorRT backend (quant_tensor.py:128-156), fp32 out.
    #    Two in / out pairs rotate (822 MB).
    n_a = 4 * n_b
    xa = [torch.cat([batches[k] for k in (0, 1, 0, 1)]).contiguous(), torch.cat([batches[k] for k in (1, 0, 1, 0)]).contiguous()]
    ya = [torch.empty_like(x) for x in xa]
    sa = s_a.reshape(1).contiguous()
    za = z_a.reshape(1).contiguous()

    def qdq_a(i):
        k = i % 2
        return lib.sbq_quant_pertensor_forward(L.ptr(xa[k]), L.F32, L.ptr(ya[k]), L.F32, None, L.Q_NONE, L.ptr(sa), L.ptr(za), n_a,
                                               -128, 127, 0, c.st)

    us = c.timed(qdq_a, 20, warm=4)
    L.check(qdq_a(0))
    torch.cuda.synchronize(c.dev)
    m = 1 << 20
    idx0 = n_b - m // 2  # a stretch that crosses the first batch boundary
    dq_ref, _ = O.qdq(xa[0].reshape(-1)[idx0:idx0 + m].cpu().numpy().reshape(1, -1), s_ref, z_ref, -128, 127, 0)
    ok_a = _same(ya[0].reshape(-1)[idx0:idx0 + m].cpu().numpy().reshape(1, -1), dq_ref)
    out["qdq_activation_256x64x56x56_fp32_per_tensor"] = _entry(
        us, n_a * 8, ok_a, "2^20 elements of the fp32 output == oracle qdq bit for bit (quant_tensor.py:128-156 at zp = 0)",
        elements=n_a)
    # the same call through Quantizer.forward (TensorRT backend): what a reference user's model pays per layer
    qa.enable_quant()
    with torch.no_grad():
        qa(xa[0])
        us_q = c.timed(lambda i: qa(xa[i % 2]), 20, warm=4)
    out["qdq_activation_256x64x56x56_fp32_per_tensor"]["through_quantizer_forward_us"] = round(us_q, 2)
    del xa, ya
    # -- (c) the largest conv weight, 512 x 512 x 3 x 3 fp32, per channel symmetric (W of the yaml)
    wshape = (512, 512, 3, 3)
    n_w = 512 * 512 * 9
    host_w = [torch.randn(wshape, generator=g) * 0.02 for _ in range(2)]
    wsd = []
    for i in range(32):  # 32 x 2 x 9.4 MB = 604 MB in rotation
        wsd.append(torch.roll(host_w[i % 2].reshape(512, -1), i // 2, 1).reshape(wshape).contiguous().to(c.dev))
    ywd = [torch.empty_like(w) for w in wsd]
    qw = build_quantizer(quantizer_config("per-channel-symmetric", 8, observer="MINMAX", target="weight"))
    qw.set_backend(Backend.TENSORRT)
    qw.update_observer(wsd[0])
    s_w, z_w = qw.calc_qparams()
    s_w = s_w.reshape(-1).contiguous()
    z_w = z_w.reshape(-1).contiguous()

    def qdq_w(i):
        k = (i % 16) * 2  # the copies that share weight 0's row statistics
        return lib.sbq_quant_perchannel_forward(L.ptr(wsd[k]), L.F32, L.ptr(ywd[k]), L.F32, None, L.Q_NONE, L.ptr(s_w), L.ptr(z_w), 1,
                                                512, 4608, -128, 127, 0, c.st)

    us = c.timed(qdq_w, 200, warm=20)
    L.check(qdq_w(0))
    torch.cuda.synchronize(c.dev)
    wf = host_w[0].numpy().reshape(512, -1)
    mn_r, mx_r = O.minmax(wf, 0, True)
    s_r, z_r = O.qparams_from_minmax(mn_r, mx_r, -128, 127, True)
    dq_ref, _ = O.qdq(wf, s_r, z_r, -128, 127, 0)
    ok_w = bool(np.array_equal(s_w.cpu().numpy(), s_r)) and _same(ywd[0].cpu().numpy().reshape(512, -1), dq_ref)
    out["qdq_weight_512x512x3x3_fp32_per_channel"] = _entry(
        us, n_w * 8, ok_w, "scale of every channel and the whole fp32 output == oracle (minmax.py:14-25, base.py:63-79, "
        "quant_tensor.py:128-156)", elements=n_w, note="9.4 MB in + 9.4 MB out: launch-latency bound (what the model-wide "
        "launch of extras.resnet50_*_weights_one_launch_us removes)")
    return out


# ------------------------------------------------------------------------------------------------------
# config 2: per-channel MSE observer
# ------------------------------------------------------------------------------------------------------
# vector instructions per candidate evaluation in mse_partial_kernel's hot loop (sbq_observe.hip) and their flops:
# mul, rndne, med3, fma, fma -> 5 instructions, 7 flops (an fma counts 2)
MSE_FLOPS_PER_EVAL = 7
# a NOMINAL issue rate for reference only -- one wave64 instruction per SIMD per 4 cycles at 2.4 GHz x 1024 SIMDs.  It is
# not a ceiling: tools/lab/valu_rate.hip measures 4.1-4.3 cycles for dependent chains of plain instructions (6.2 for
# v_pk_fma_f32), the MSE kernel with four independent waves per SIMD sustains ~3.4.  The roof to judge by is the
# 157.3 TFLOP/s fp32 vector peak.
VALU_ISSUE_CEILING_WAVE_INSTS = 2.4e9 / 4 * 1024


def config2_mse(c):
    from oracle import oracle as O

    L, lib, ops = c.L, c.lib, c.ops
    x = c.xs[0]
    mn, mx, _ = ops.channel_stats(x, 0, True)
    sse = torch.zeros(ROWS, L.MSE_CANDIDATES, dtype=torch.float64, device=c.dev)
    ws = torch.empty(max(lib.sbq_mse_workspace_bytes(1, ROWS, COLS), 16), dtype=torch.uint8, device=c.dev)
    s_o = torch.empty(ROWS, dtype=torch.float32, device=c.dev)
    z_o = torch.empty(ROWS, dtype=torch.float32, device=c.dev)
    idx = torch.empty(ROWS, dtype=torch.int32, device=c.dev)

    def acc(i):
        lib.sbq_mse_accumulate(L.ptr(c.xs[i % len(c.xs)]), L.BF16, 1, ROWS, COLS, L.ptr(mn), L.ptr(mx), -128, 127, 1,
                               L.ptr(sse), L.ptr(ws), ws.numel(), c.st)

    def whole(i):
        # the observer's calc_qparams as the library runs it: statistics, clear, accumulate, select
        xi = c.xs[i % len(c.xs)]
        lib.sbq_channel_stats(L.ptr(xi), L.BF16, 1, ROWS, COLS, L.ptr(mn), L.ptr(mx), None, L.ptr(ws), ws.numel(), c.st)
        sse.zero_()
        lib.sbq_mse_accumulate(L.ptr(xi), L.BF16, 1, ROWS, COLS, L.ptr(mn), L.ptr(mx), -128, 127, 1, L.ptr(sse), L.ptr(ws),
                               ws.numel(), c.st)
        lib.sbq_mse_select(L.ptr(sse), ctypes.c_double(COLS), L.ptr(mn), L.ptr(mx), ROWS, -128, 127, 1, L.ptr(s_o), L.ptr(z_o),
                           L.ptr(idx), c.st)

    k_us = c.timed(acc, 20, warm=3)
    e2e_us = c.timed(whole, 20, warm=3)
    # gate: argmin index of 128 rows against the oracle (a differing index passes only when the oracle's own fp64
    # losses of the two candidates agree to 1e-7: a tie the fp32 reference itself resolves by summation order)
    whole(0)
    torch.cuda.synchronize(c.dev)
    rows = list(range(0, ROWS, 32))
    _, _, b_ref, sse_ref = O.mse(c.host.float().numpy()[rows], -128, 127, True, 0, True)
    b_gpu = idx[rows].cpu().numpy()
    ok = True
    n_diff = 0
    for r in range(len(rows)):
        if b_gpu[r] != b_ref[r]:
            n_diff += 1
            a, b = sse_ref[r, b_gpu[r]], sse_ref[r, b_ref[r]]
            ok &= abs(a - b) <= 1e-7 * max(abs(a), abs(b))
    evals = c.n * L.MSE_CANDIDATES
    tflops = evals * MSE_FLOPS_PER_EVAL / k_us / 1e6
    # the same tensor taken as a WHOLE (a per-tensor MSE quantizer: config 2's activations when the model runs in
    # bf16): 16-bit values -> an exact histogram, 80 candidates x 65 536 values instead of 80 x 16.7 M elements
    mn1, mx1, _ = ops.channel_stats(x, 0, False)
    sse1 = torch.zeros(1, L.MSE_CANDIDATES, dtype=torch.float64, device=c.dev)
    ws1 = torch.empty(max(lib.sbq_mse_workspace_bytes(1, 1, c.n), 16), dtype=torch.uint8, device=c.dev)

    def acc1(i):
        lib.sbq_mse_accumulate(L.ptr(c.xs[i % len(c.xs)]), L.BF16, 1, 1, c.n, L.ptr(mn1), L.ptr(mx1), -128, 127, 1, L.ptr(sse1),
                               L.ptr(ws1), ws1.numel(), c.st)

    t_us = c.timed(acc1, 20, warm=3)
    L.set_tuning(2, 19)
    try:
        t_old_us = c.timed(acc1, 10, warm=2)
    finally:
        L.set_tuning(2, 0)
    sse1.zero_()
    acc1(0)
    torch.cuda.synchronize(c.dev)
    _, _, b1_ref, sse1_ref = O.mse(c.host.float().numpy().reshape(1, -1), -128, 127, True, 0, False)
    got1 = sse1.cpu().numpy()[0]
    ok1 = bool(np.all(np.abs(got1 - sse1_ref[0]) <= 1e-5 * np.abs(sse1_ref[0])) and int(np.argmin((got1 / c.n).astype(np.float32))) == int(b1_ref[0]))
    per_tensor = _entry(t_us, c.n * 2, ok1, "all 80 squared-error sums within 1e-5 of the oracle's fp64 sums and the same argmin "
                        "(observers/mse.py:46-61 per tensor)", bound="hbm", launches=4,
                        per_element_route_us=round(t_old_us, 2), note="clear + histogram + 80 x 65 536 evaluation + fold")
    return _entry(
        k_us, c.n * 2, ok,
        "argmin candidate index of %d rows == oracle (observers/mse.py:51-61); %d differing" % (len(rows), n_diff),
        bound="valu",
        end_to_end_us=round(e2e_us, 2),
        candidate_evaluations=evals,
        valu_tflops=round(tflops, 1),
        frac_of_fp32_vector_peak=round(tflops / FP32_VECTOR_PEAK_TFLOPS, 4),
        vs_nominal_issue_rate_of_1_per_4_cycles=round(evals * 5 / 64 / (k_us * 1e-6) / VALU_ISSUE_CEILING_WAVE_INSTS, 4),
        per_tensor_histogram_route=per_tensor,
        note="VALU-bound: x is read once (2 B/elem); frac is the HBM fraction of that read, the roof that matters is "
             "frac_of_fp32_vector_peak (7 flops per candidate evaluation / 157.3 TFLOP/s); the nominal issue rate (5 wave "
             "instructions per 64 evaluations against one instruction per SIMD per 4 cycles) is a yardstick the kernel exceeds",
    )


# ------------------------------------------------------------------------------------------------------
# config 3: percentile observer over the cached calibration batches (DeiT-small), per tensor
# ------------------------------------------------------------------------------------------------------
def config3_percentile(c):
    from oracle import oracle as O

    L, lib, ops = c.L, c.lib, c.ops
    g = torch.Generator().manual_seed(33)
    sets = []
    # 3 x 4 batches: the calibration set rotates so that a call does not find its data in the Infinity Cache
    for _ in range(3):
        batches = []
        for _ in range(4):
            a = torch.randn(64, 197, 384, generator=g)
            a = a * (1.0 + 9.0 * (torch.rand(1, 1, 384, generator=g) > 0.97))  # a few outlier channels, as ViT activations have
            batches.append(a.bfloat16())
        sets.append(batches)
    dsets = [[b.to(c.dev) for b in s] for s in sets]
    n = sum(b.numel() for b in sets[0])

    def run(i):
        return ops.percentile_select(dsets[i % 3], 1e-3, 0, False)

    us = c.timed(run, 30, warm=5)
    mn, mx = run(0)
    torch.cuda.synchronize(c.dev)
    flat = np.concatenate([_bf16_np(b).reshape(-1) for b in sets[0]])
    mn_ref, mx_ref = O.percentile(flat, 1e-3, 0, False)
    ok = bool(np.array_equal(mn.cpu().numpy().reshape(-1), mn_ref) and np.array_equal(mx.cpu().numpy().reshape(-1), mx_ref))
    out = {
        "deit_4_batches_per_tensor": _entry(us, n * 2, ok, "(min, max) of 4 x 64x197x384 bf16 batches bit-exact vs oracle "
                                            "(percentile.py:16-46 on the concatenated data)", elements=n),
    }
    # the headline tensor: per tensor and per channel
    w = c.xs[0]
    us_t = c.timed(lambda i: ops.percentile_select([c.xs[i % len(c.xs)]], 1e-3, 0, False), 30, warm=5)
    mn, mx = ops.percentile_select([w], 1e-3, 0, False)
    mn_ref, mx_ref = O.percentile(c.host.float().numpy().reshape(-1), 1e-3, 0, False)
    ok = bool(np.array_equal(mn.cpu().numpy().reshape(-1), mn_ref) and np.array_equal(mx.cpu().numpy().reshape(-1), mx_ref))
    out["weight_per_tensor"] = _entry(us_t, c.n * 2, ok, "(min, max) of the 4096x4096 weight bit-exact vs oracle")
    us_c = c.timed(lambda i: ops.percentile_rows(c.xs[i % len(c.xs)], 1e-3), 30, warm=5)
    mn, mx = ops.percentile_rows(w, 1e-3)
    rows = GATE_ROWS
    mn_ref, mx_ref = O.percentile(c.host.float().numpy()[rows], 1e-3, 0, True)
    ok = bool(np.array_equal(mn[rows].cpu().numpy(), mn_ref) and np.array_equal(mx[rows].cpu().numpy(), mx_ref))
    out["weight_per_channel"] = _entry(us_c, c.n * 2, ok, "(min, max) of %d rows bit-exact vs oracle" % len(rows))
    # the same four batches as fp32 -- what a reference user's fp32 model caches (round 6: ONE launch, the keys inside the
    # first windows kept in LDS, instead of three launches that each sweep every batch)
    fsets = [[b.float() for b in s] for s in dsets]
    us_f = c.timed(lambda i: ops.percentile_select(fsets[i % 3], 1e-3, 0, False), 20, warm=4)
    mn, mx = ops.percentile_select(fsets[0], 1e-3, 0, False)
    torch.cuda.synchronize(c.dev)
    mn_ref, mx_ref = O.percentile(flat, 1e-3, 0, False)  # (the fp32 values ARE the bf16 ones)
    ok = bool(np.array_equal(mn.cpu().numpy().reshape(-1), mn_ref) and np.array_equal(mx.cpu().numpy().reshape(-1), mx_ref))
    out["deit_4_batches_per_tensor_fp32"] = _entry(us_f, n * 4, ok, "(min, max) of the same 4 batches held as fp32, bit-exact vs "
                                                   "oracle; one launch (candidate store in LDS)", elements=n)
    del fsets
    return out


# ------------------------------------------------------------------------------------------------------
# config 4: GPTQ 4-bit group-128 mat-vec, B = 1, the three LLaMA-7B linear shapes
# ------------------------------------------------------------------------------------------------------
def _gptq_problem(in_f, out_f, seed, dev, copies):
    g = torch.Generator().manual_seed(seed)
    groups = in_f // 128
    qws, scs, zrs = [], [], []
    for _ in range(copies):
        qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (in_f // 8, out_f), generator=g, dtype=torch.int64).to(torch.int32)
        sc = (torch.rand(out_f, groups, generator=g) * 0.02 + 0.001).float()
        zr = (torch.randint(0, 16, (out_f, groups), generator=g).float() * sc).float()  # zeros' = zero * scale (quant.py:205)
        qws.append(qw.to(dev))
        scs.append(sc.to(dev))
        zrs.append(zr.to(dev))
    x = torch.randn(1, in_f, generator=g).float()
    return qws, scs, zrs, x


def config4_gptq(c):
    from oracle import oracle as O

    L, lib = c.L, c.lib
    out = {}
    for name, in_f, out_f in (("4096x4096", 4096, 4096), ("4096->11008", 4096, 11008), ("11008->4096", 11008, 4096)):
        groups = in_f // 128
        w_bytes = in_f // 8 * out_f * 4
        copies = max(2, int(3.2e8 // w_bytes) + 1)  # > 256 MiB of weights in rotation: every call reads HBM
        qws, scs, zrs, x = _gptq_problem(in_f, out_f, 4 + in_f % 97, c.dev, copies)
        xd = x.to(c.dev)
        y = torch.zeros(1, out_f, dtype=torch.float32, device=c.dev)
        ws = L.fresh_workspace(max(lib.sbq_gptq_workspace_bytes(1, in_f, out_f), 16), c.dev)
        args = [(L.ptr(xd), L.ptr(qws[j]), L.ptr(y), L.ptr(scs[j]), L.ptr(zrs[j])) for j in range(copies)]

        def run(i):
            a = args[i % copies]
            lib.sbq_vecquant4matmul(a[0], a[1], a[2], a[3], a[4], 1, in_f, out_f, 128, L.ptr(ws), ws.numel(), c.st)

        us = c.timed(run, 400, warm=40)
        us_warm = c.timed(lambda i: run(0), 400, warm=40)
        # gate: the reference's own test criterion (test_cuda_kernel.py:21-126: rtol = atol = 1e-5 against the
        # dequantized-weight product), through the oracle's restatement of cuda_kernel_4bit.cu
        y.zero_()
        run(0)
        torch.cuda.synchronize(c.dev)
        ref = O.vecquantmatmul(x.numpy(), qws[0].cpu().numpy(), np.zeros(out_f, np.float32), scs[0].cpu().numpy(),
                               zrs[0].cpu().numpy(), 128, 4)
        got = y.cpu().numpy()
        # the reference's criterion, literally: torch.allclose(rtol = 1e-5, atol = 1e-5)  (test_cuda_kernel.py:47)
        ok = bool(np.all(np.abs(got - ref) <= 1e-5 + 1e-5 * np.abs(ref)))
        nbytes = w_bytes + 2 * out_f * groups * 4 + (in_f + 2 * out_f) * 4
        out[name] = _entry(us, nbytes, ok, "y == oracle(cuda_kernel_4bit.cu:36-180) at the reference test's literal rtol = atol "
                           "= 1e-5 (test_cuda_kernel.py:47)", weight_copies_in_rotation=copies,
                           cache_resident_us=round(us_warm, 3), max_abs_err=float(np.abs(got - ref).max()))
        # ---- the same matrix at B = 8 and B = 32 (SURVEY.md 8(d) M-gptq: B in {1, 8, 32}; the reference's multi-batch
        #      cases, test_cuda_kernel.py:81-126).  The weight stream is read once per call whatever B is, so the HBM
        #      fraction falls with B while the arithmetic grows (2 * B * in * out flops in true fp32 -- the contract is fp32
        #      FMA on int nibbles; B >= 5 runs on the fp32 matrix cores, gptq_mfma_kernel: v_mfma_f32_16x16x4_f32, whose
        #      peak equals the fp32 vector peak, 157.3 TFLOP/s): both fractions are reported.
        gx = torch.Generator().manual_seed(900 + in_f % 89)
        for B in (8, 32):
            xb = torch.randn(B, in_f, generator=gx).float()
            xbd = xb.to(c.dev)
            yb = torch.zeros(B, out_f, dtype=torch.float32, device=c.dev)
            wsb = L.fresh_workspace(max(lib.sbq_gptq_workspace_bytes(B, in_f, out_f), 16), c.dev)
            argb = [(L.ptr(xbd), L.ptr(qws[j]), L.ptr(yb), L.ptr(scs[j]), L.ptr(zrs[j])) for j in range(copies)]

            def run_b(i):
                a = argb[i % copies]
                return lib.sbq_vecquant4matmul(a[0], a[1], a[2], a[3], a[4], B, in_f, out_f, 128, L.ptr(wsb), wsb.numel(), c.st)

            us_b = c.timed(run_b, 200, warm=20)
            yb.zero_()
            L.check(run_b(0))
            torch.cuda.synchronize(c.dev)
            ref = O.vecquantmatmul(xb.numpy(), qws[0].cpu().numpy(), np.zeros(out_f, np.float32), scs[0].cpu().numpy(),
                                   zrs[0].cpu().numpy(), 128, 4)
            got = yb.cpu().numpy()
            ok_b = bool(np.all(np.abs(got - ref) <= 1e-5 + 1e-5 * np.abs(ref)))  # (literal: test_cuda_kernel.py:47)
            nbytes_b = w_bytes + 2 * out_f * groups * 4 + B * (in_f + 2 * out_f) * 4
            flops = 2.0 * B * in_f * out_f
            out.setdefault("B%d" % B, {})[name] = _entry(
                us_b, nbytes_b, ok_b, "y [%d, %d] == oracle(cuda_kernel_4bit.cu:36-180) at the reference test's literal rtol = atol = "
                "1e-5" % (B, out_f), batch=B, weight_copies_in_rotation=copies, max_abs_err=float(np.abs(got - ref).max()),
                kernel="gptq_mfma_kernel (fp32 matrix cores)", fp32_tflops=round(flops / us_b / 1e6, 2),
                frac_of_fp32_matrix_peak=round(flops / us_b / 1e6 / FP32_VECTOR_PEAK_TFLOPS, 4),
                us_per_row_of_x=round(us_b / B, 3))
            del xbd, yb, wsb
        del qws, scs, zrs
    # ---- a decoder layer's projections that share their input, as ONE launch each (quant.py:262-278 issues one
    #      mat-vec per QuantLinear): q / k / v = 3 x (4096 -> 4096); gate + up = 2 x (4096 -> 11008)
    for name, in_f, outs in (("qkv_3x4096x4096_one_launch", 4096, (4096, 4096, 4096)),
                             ("gate_up_2x4096x11008_one_launch", 4096, (11008, 11008))):
        groups = in_f // 128
        w_bytes = sum(in_f // 8 * o * 4 for o in outs)
        copies = max(2, int(3.2e8 // w_bytes) + 1)
        sets = []
        for cpy in range(copies):
            mats = []
            for m, o in enumerate(outs):
                qws, scs, zrs, x = _gptq_problem(in_f, o, 40 + cpy * 8 + m, c.dev, 1)
                mats.append((qws[0], scs[0], zrs[0]))
            sets.append(mats)
        xd = x.to(c.dev)
        ys = [torch.zeros(1, o, dtype=torch.float32, device=c.dev) for o in outs]
        total = sum(outs)
        ws = L.fresh_workspace(max(lib.sbq_gptq_workspace_bytes(1, in_f, total), 16), c.dev)
        n = len(outs)
        arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])  # noqa: E731
        outf = (ctypes.c_int64 * n)(*outs)
        yp = arr(ys)
        margs = [(arr([m[0] for m in mats]), arr([m[1] for m in mats]), arr([m[2] for m in mats])) for mats in sets]

        def run(i):
            a = margs[i % copies]
            return lib.sbq_vecquantmatmul_multi(4, L.ptr(xd), n, a[0], yp, a[1], a[2], outf, 1, in_f, 128, L.ptr(ws), ws.numel(), c.st)

        us = c.timed(run, 300, warm=30)

        def run_single(i):
            mats = sets[i % copies]
            for m in range(n):
                lib.sbq_vecquant4matmul(L.ptr(xd), L.ptr(mats[m][0]), L.ptr(ys[m]), L.ptr(mats[m][1]), L.ptr(mats[m][2]), 1, in_f,
                                        outs[m], 128, L.ptr(ws), ws.numel(), c.st)

        us_1 = c.timed(run_single, 200, warm=20)
        for y in ys:
            y.zero_()
        L.check(run(0))
        torch.cuda.synchronize(c.dev)
        ok = True
        worst = 0.0
        for m in range(n):
            ref = O.vecquantmatmul(x.numpy(), sets[0][m][0].cpu().numpy(), np.zeros(outs[m], np.float32), sets[0][m][1].cpu().numpy(),
                                   sets[0][m][2].cpu().numpy(), 128, 4)
            got = ys[m].cpu().numpy()
            ok = ok and bool(np.all(np.abs(got - ref) <= 1e-5 + 1e-5 * np.abs(ref)))  # (literal: test_cuda_kernel.py:47)
            worst = max(worst, float(np.abs(got - ref).max()))
        nbytes = w_bytes + sum(2 * o * groups * 4 for o in outs) + (in_f + 2 * total) * 4
        out[name] = _entry(us, nbytes, ok, "every matrix's y == oracle at the reference test's literal rtol = atol = 1e-5", launches=1,
                           one_launch_per_matrix_us=round(us_1, 3), us_per_4096x4096_equivalent=round(us * 9486336 / nbytes, 3),
                           weight_copies_in_rotation=copies, max_abs_err=worst)
    return out


# ------------------------------------------------------------------------------------------------------
# config 5: 50 % unstructured mask + LSQ 4-bit: threshold, mask, fused forward, STE backward
# ------------------------------------------------------------------------------------------------------
def config5_mask_lsq(c):
    from oracle import oracle as O

    L, lib, ops = c.L, c.lib, c.ops
    nb = len(c.xs)
    # a conv / linear weight as config 5 has them (ResNet-50: rows of one magnitude).  The headline weight's rows span
    # three decades: a GLOBAL 50 % threshold prunes its small rows completely, LSQ's init then gives them scale 0 and
    # both the reference and the kernel return NaN there -- a degenerate workload, not a timing one.
    g5 = torch.Generator().manual_seed(5)
    host5 = (torch.randn(ROWS, COLS, generator=g5) * 0.05).bfloat16()
    xs5 = [host5.to(c.dev)]
    for j in range(1, nb):
        xs5.append(torch.roll(xs5[0], shifts=j, dims=1).contiguous())
    c = _With(c, xs=xs5, host=host5)
    w = c.xs[0]
    wf = c.host.float().numpy()
    out = {}
    # -- threshold: k-th smallest |w| (l1norm.py:21-23), the whole tensor against the oracle's sort
    idx = min(int(c.n * 0.5), c.n - 1)
    mask_ref, thr_ref = O.l1_mask(wf, 0.5)
    ws = L.fresh_workspace(max(lib.sbq_radix_select_workspace_bytes(1, 1), 16), c.dev)
    thr = torch.empty((), dtype=torch.float32, device=c.dev)

    def kth(i):
        lib.sbq_kth_value(L.ptr(c.xs[i % nb]), L.BF16, c.n, 1, idx + 1, L.ptr(thr), L.ptr(ws), ws.numel(), c.st)

    us = c.timed(kth, 100)
    kth(0)
    torch.cuda.synchronize(c.dev)
    out["mask_threshold_kth_value"] = _entry(us, c.n * 2, float(thr.item()) == float(thr_ref),
                                             "threshold == sort(|w|)[n/2] of the oracle, exact")
    # the same weight held as fp32 -- a reference user's fp32 model (round 6: one launch instead of three)
    xf = [c.xs[i].float() for i in range(min(nb, 4))]
    thr_f = torch.empty((), dtype=torch.float32, device=c.dev)

    def kth_f(i):
        lib.sbq_kth_value(L.ptr(xf[i % len(xf)]), L.F32, c.n, 1, idx + 1, L.ptr(thr_f), L.ptr(ws), ws.numel(), c.st)

    us_f = c.timed(kth_f, 50)
    kth_f(0)
    torch.cuda.synchronize(c.dev)
    out["mask_threshold_kth_value_fp32"] = _entry(us_f, c.n * 4, float(thr_f.item()) == float(thr_ref),
                                                  "the same weight as fp32: threshold == sort(|w|)[n/2] of the oracle, exact; one "
                                                  "launch (candidate store in LDS)")
    del xf
    # -- mask = |w| > thresh (l1norm.py:24-25): all 16.7 M bytes against the oracle's mask
    masks = [torch.empty(ROWS, COLS, dtype=torch.uint8, device=c.dev) for _ in range(nb)]
    us = c.timed(lambda i: lib.sbq_mask_from_threshold(L.ptr(c.xs[i % nb]), L.BF16, c.n, L.ptr(thr), L.ptr(masks[i % nb]), c.st), 100)
    lib.sbq_mask_from_threshold(L.ptr(w), L.BF16, c.n, L.ptr(thr), L.ptr(masks[0]), c.st)
    torch.cuda.synchronize(c.dev)
    ok = bool(np.array_equal(masks[0].cpu().numpy().astype(bool), mask_ref))
    out["mask_from_threshold"] = _entry(us, c.n * 3, ok, "all 16.7 M mask bytes == oracle (strict >, ties pruned)")
    for j in range(1, nb):  # masks of the rotated copies (same values, rolled)
        lib.sbq_mask_from_threshold(L.ptr(c.xs[j]), L.BF16, c.n, L.ptr(thr), L.ptr(masks[j]), c.st)
    # -- LSQ 4-bit scales of the masked weight (lsq.py:44-47 on w * mask), from the oracle so that the gate below
    #    isolates the forward kernel
    rows = GATE_ROWS
    s_all = O.lsq_init_scale(wf * mask_ref, 7, 0, True)
    s_raw = torch.from_numpy(-s_all).to(c.dev)  # negative raw parameter: the kernel applies |s| (lsq.py:61)
    z_raw = torch.zeros(ROWS, dtype=torch.float32, device=c.dev)
    dq_ref, _ = O.qdq(wf[rows], s_all[rows], np.zeros(len(rows), np.float32), -8, 7, 0, mask=mask_ref[rows])
    want = _as_bf16_bits(dq_ref)

    def fused_bytes(i):
        j = i % nb
        return lib.sbq_quant_lsq_forward(L.ptr(c.xs[j]), L.BF16, L.ptr(c.ys[j]), L.BF16, L.ptr(masks[j]), L.ptr(s_raw),
                                         L.ptr(z_raw), 1, ROWS, COLS, -8, 7, c.st)

    us = c.timed(fused_bytes, 200)
    c.ys[0].zero_()
    L.check(fused_bytes(0))
    torch.cuda.synchronize(c.dev)
    ok = _same(c.ys[0][rows].float().cpu().numpy(), want)
    out["fused_mask_bytes_lsq_qdq"] = _entry(us, c.n * 5, ok, "bf16 output of %d rows == RNE(oracle qdq(w * mask)) "
                                             "(sparse/modules/conv.py:39-43 + lsq.py:61-76), exact" % len(rows))
    s_abs = torch.from_numpy(s_all).to(c.dev)

    def fused_thr(i):
        j = i % nb
        return lib.sbq_mask_quant_forward(L.ptr(c.xs[j]), L.BF16, L.ptr(c.ys[j]), L.BF16, None, L.Q_NONE, None, L.ptr(thr),
                                          L.ptr(s_abs), L.ptr(z_raw), 1, ROWS, COLS, -8, 7, 0, c.st)

    us = c.timed(fused_thr, 200)
    c.ys[0].zero_()
    L.check(fused_thr(0))
    torch.cuda.synchronize(c.dev)
    ok = _same(c.ys[0][rows].float().cpu().numpy(), want)
    out["fused_threshold_qdq"] = _entry(us, c.n * 4, ok, "same rows, mask recomputed from the threshold in the kernel (no mask bytes)")
    # -- STE backward with the LSQ step-size gradient (fake_quant_tensor.cu:227-270 + lsq.py:13-21)
    g = torch.Generator().manual_seed(55)
    gy_h = torch.randn(ROWS, COLS, generator=g).bfloat16()
    gys = [gy_h.to(c.dev)]
    for j in range(1, nb):
        gys.append(torch.roll(gys[0], shifts=j, dims=1).contiguous())
    gxs = [torch.empty(ROWS, COLS, dtype=torch.bfloat16, device=c.dev) for _ in range(nb)]
    gs = torch.empty(ROWS, dtype=torch.float32, device=c.dev)
    bws = torch.empty(max(lib.sbq_backward_workspace_bytes(1, ROWS, COLS), 16), dtype=torch.uint8, device=c.dev)
    ratio = 1.0 / math.sqrt(COLS * 7)

    def bwd(i):
        j = i % nb
        return lib.sbq_quant_lsq_backward(L.ptr(c.xs[j]), L.ptr(gys[j]), L.BF16, L.ptr(gxs[j]), L.BF16, L.ptr(gs), L.ptr(s_raw),
                                          L.ptr(z_raw), 1, ROWS, COLS, -8, 7, ctypes.c_float(ratio), L.ptr(bws), bws.numel(), c.st)

    us = c.timed(bwd, 100)
    L.check(bwd(0))
    torch.cuda.synchronize(c.dev)
    gx_ref, gs_ref, _ = O.ste_backward(wf[rows], gy_h.float().numpy()[rows], s_all[rows], np.zeros(len(rows), np.float32), -8, 7, 0)
    ok_gx = _same(gxs[0][rows].float().cpu().numpy(), _as_bf16_bits(gx_ref))
    gs_want = gs_ref.astype(np.float64) * ratio * -1.0  # sign(raw scale) = -1
    gs_got = gs[rows].cpu().numpy().astype(np.float64)
    rel = float(np.max(np.abs(gs_got - gs_want) / np.maximum(np.abs(gs_want), 1e-12)))
    out["ste_backward_with_gs"] = _entry(us, c.n * 6, ok_gx and rel <= 1e-5,
                                         "gx of %d rows exact vs oracle (MySTE.backward, quant_tensor.py:45-71); "
                                         "gs * gs_ratio * sign(s) within 1e-5 relative (fp32 summation order is free)" % len(rows),
                                         gs_max_rel_err=rel)
    return out


# ------------------------------------------------------------------------------------------------------
# model-wide calibration launches: ResNet-50's 53 conv / fc weights (fp32 masters), one by one vs grouped
#   tools/calibration.py:117-135 (weight calibration loops the layers), sparse/sparse_model.py:107-113 (mask thresholds)
# ------------------------------------------------------------------------------------------------------
def resnet50_weight_shapes():
    shapes = []
    inp = 64
    for width, blocks in ((64, 3), (128, 4), (256, 6), (512, 3)):
        for b in range(blocks):
            shapes += [(width, inp, 1, 1), (width, width, 3, 3), (width * 4, width, 1, 1)]
            if b == 0:
                shapes.append((width * 4, inp, 1, 1))
            inp = width * 4
    shapes.append((1000, 2048))
    return shapes


def model_wide_calibration(c):
    from oracle import oracle as O

    L, lib, ops = c.L, c.lib, c.ops
    g = torch.Generator().manual_seed(50)
    ws = [torch.randn(shp, generator=g).to(c.dev) for shp in resnet50_weight_shapes()]
    n_elem = sum(w.numel() for w in ws)
    nbytes = n_elem * 4
    out = {"tensors": len(ws), "elements": n_elem}

    def sync_time(fn, iters):
        return c.timed(lambda i: fn(), iters, warm=3)

    # ---- min-max observer + calc_qparams (8-bit per channel symmetric) ----
    grp = ops.GroupCalibration([(w, -128, 127, True, True) for w in ws])

    def one_by_one_minmax():
        res = []
        for w in ws:
            mn, mx, _ = ops.channel_stats(w, 0, True)
            res.append((mn, mx) + tuple(ops.qparams_from_minmax(mn, mx, -128, 127, True)))
        return res

    ref = one_by_one_minmax()
    mn_g, mx_g, s_g, z_g = grp.minmax_qparams()
    torch.cuda.synchronize(c.dev)
    same = all(torch.equal(a[0], mn_g[i]) and torch.equal(a[1], mx_g[i]) and torch.equal(a[2].reshape(-1), s_g[i]) and
               torch.equal(a[3].reshape(-1), z_g[i]) for i, a in enumerate(ref))
    k = 7  # and against the oracle on one tensor
    mn_o, mx_o = O.minmax(ws[k].cpu().numpy().reshape(ws[k].shape[0], -1), 0, True)
    s_o, _ = O.qparams_from_minmax(mn_o, mx_o, -128, 127, True)
    same = same and bool(np.array_equal(mn_g[k].cpu().numpy(), mn_o) and np.array_equal(s_g[k].cpu().numpy(), s_o))
    us_1 = sync_time(one_by_one_minmax, 5)
    us_g = sync_time(grp.launch_minmax, 50)
    out["minmax_qparams"] = _entry(us_g, nbytes, same, "grouped (2 launches) == per-tensor calls bit for bit on all %d tensors; "
                                   "tensor %d == oracle" % (len(ws), k), one_by_one_us=round(us_1, 1), launches_grouped=2,
                                   launches_one_by_one=2 * len(ws))
    # ---- MSE observer (per channel, 80 candidates) ----
    def one_by_one_mse():
        res = []
        for w in ws:
            mn, mx, _ = ops.channel_stats(w, 0, True)
            sse = torch.zeros(w.shape[0], L.MSE_CANDIDATES, dtype=torch.float64, device=c.dev)
            ops.mse_accumulate(w, mn, mx, -128, 127, True, sse, 0, True)
            res.append(ops.mse_select(sse, w[0].numel(), mn, mx, -128, 127, True))
        return res

    ref = one_by_one_mse()
    s_m, z_m, i_m = grp.mse_qparams()
    torch.cuda.synchronize(c.dev)
    # (round 6: the grouped launch has a summation tree of its own; rows whose index differs from the per-tensor kernel's
    # must be ties of the two candidates' losses in the oracle's fp64 sums -- oracle.mse_index_disagreements)
    same, differing = True, 0
    for i, a in enumerate(ref):
        eq = a[2].reshape(-1) == i_m[i]
        n_diff = int((~eq).sum())
        differing += n_diff
        same = same and torch.equal(a[0].reshape(-1)[eq], s_m[i][eq])
        if n_diff:
            rows = ws[i].reshape(ws[i].shape[0], -1).cpu().numpy()
            same = same and O.mse_index_disagreements(rows, i_m[i].cpu().numpy(), a[2].reshape(-1).cpu().numpy(), -128, 127, True) == []
    rows_k = ws[k].cpu().numpy().reshape(ws[k].shape[0], -1)[:64]
    _, _, b_o, _ = O.mse(rows_k, -128, 127, True, 0, True)
    same = same and O.mse_index_disagreements(rows_k, i_m[k][:64].cpu().numpy(), b_o, -128, 127, True) == []
    us_1 = sync_time(one_by_one_mse, 3)
    us_g = sync_time(grp.launch_mse, 20)
    evals = n_elem * L.MSE_CANDIDATES
    out["mse_qparams"] = _entry(us_g, nbytes, same, "grouped (3 launches: two for min-max, one for the search and the pick; a lane per (row, candidate)) vs per-tensor calls: argmin index "
                                "and scale of every channel of all %d tensors -- equal, or the two candidates' losses tie within 2e-6 in "
                                "the oracle's fp64 sums (%d such rows); 64 rows of tensor %d vs the oracle's index likewise"
                                % (len(ws), differing, k),
                                bound="valu", one_by_one_us=round(us_1, 1), launches_grouped=3,
                                valu_tflops=round(evals * MSE_FLOPS_PER_EVAL / us_g / 1e6, 1))
    # ---- L1 mask thresholds at 50 % ----
    ks = [min(int(w.numel() * 0.5), w.numel() - 1) + 1 for w in ws]

    def one_by_one_kth():
        return [ops.kth_value(w, kk, True) for w, kk in zip(ws, ks)]

    ref = torch.stack(one_by_one_kth())
    got = ops.group_kth_value(ws, ks, True)
    torch.cuda.synchronize(c.dev)
    same = bool(torch.equal(ref, got))
    for j in (0, 7, len(ws) - 1):
        a = np.abs(ws[j].cpu().numpy().reshape(-1))
        same = same and float(got[j].item()) == float(np.partition(a, ks[j] - 1)[ks[j] - 1])
    us_1 = sync_time(one_by_one_kth, 5)
    us_g = sync_time(lambda: ops.group_kth_value(ws, ks, True), 30)
    out["mask_thresholds"] = _entry(us_g, nbytes, same, "grouped (2 launches: the first collects the keys inside each first window, the second -- resident -- finishes on those) == per-tensor sbq_kth_value on all %d "
                                    "tensors; three of them == numpy partition" % len(ws), one_by_one_us=round(us_1, 1),
                                    launches_grouped=2)
    return out


# ------------------------------------------------------------------------------------------------------
# end to end: a quantized ResNet-20 forward (batch 16) with the host in and out of the way
# ------------------------------------------------------------------------------------------------------
def e2e_resnet20(c):
    """62 quantizer calls per forward around 22 convolutions / 1 linear / 9 adds (examples/resnet20_quantopr.py, the
    reference's QuantOpr convention: modules/conv.py:37-42 -> quantizers/base.py:55-64).  Host wall clock per forward:
    the generic Python route per call (what round 4 shipped), launch plans (sparsebit_amd.plan), one captured hipGraph
    (sparsebit_amd.graph); outputs compared bit for bit."""
    from examples import resnet20_quantopr as R

    rec = R.measure(c.dev, iters=200)
    rec["parity"] = bool(rec["plan_equals_eager"] and rec["graph_equals_eager"])
    rec["gate"] = "logits of the planned and of the replayed forward == the generic route's, bit for bit"
    return rec
