"""Round-5 changes on the device:
  * the 16-bit MSE histogram route with a tensor whose LAST library launch holds fewer than 8 elements (ADVICE r04:
    n_packs == 0 made the address clamp wrap) and with a one-key workgroup next to a ragged tail of other keys,
  * the windowed whole-tensor percentile with MORE cached batches than one library call takes (the exchange sequence
    must not depend on a rank's batch count),
  * per-thread tuning knobs with a process-wide default (autograd's backward threads),
  * the launch plan of a calibrated quantizer and the hipGraph capture of a whole quantized forward
    (sparsebit_amd.graph, also with the weight quantizers frozen into the capture): bit-identical to the eager path,
    invalidated by a re-calibration; a shared quantizer's several input signatures,
  * the GPTQ 4-bit batched mat-mul on the fp32 matrix cores (gptq_mfma_kernel, B >= 5) against the oracle and the strip
    tiles it replaces, determinism, inf / NaN rows.
Reference anchors: observers/mse.py:46-61, observers/percentile.py:16-46, quantizers/base.py:55-64, modules/conv.py:37-42,
large_language_models/llama/quantization/cuda/cuda_kernel_4bit.cu:36-180, test_cuda_kernel.py:81-126.
"""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from sparsebit_amd import ops as _ops

    return _ops


@pytest.fixture(scope="module")
def oracle_mod():
    from oracle import oracle as O

    return O


def _tables(ops, x, qmin, qmax, sym):
    from sparsebit_amd import lib as L

    dev = x.device
    mn, mx, _ = ops.channel_stats(x, 0, False)
    out = {}
    for knob in (0, 19):  # 19: the per-element route
        L.set_tuning(2, knob)
        try:
            sse = torch.zeros(1, L.MSE_CANDIDATES, dtype=torch.float64, device=dev)
            ops.mse_accumulate(x, mn, mx, qmin, qmax, sym, sse, 0, False)
            out[knob] = sse.cpu().numpy()[0]
        finally:
            L.set_tuning(2, 0)
    return out


@pytest.mark.parametrize("extra", [1, 5, 7, 8, 9])
def test_mse_histogram_route_short_last_launch(ops, extra):
    """numel = (65 536 x CUs) + extra: a rest of 1..7 elements used to become a launch of its own with no whole pack in it"""
    dev = torch.device("cuda:0")
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    g = torch.Generator().manual_seed(5 + extra)
    n = 65536 * cus + extra
    x = (torch.randn(n, generator=g) * 0.5).bfloat16()
    x[-extra:] = torch.tensor([3.0, -2.5, 1.75, 0.0, -0.0, 2.25, -3.5, 0.5, 4.0][:extra]).bfloat16()  # values only the rest holds
    t = _tables(ops, x.to(dev), -128, 127, True)
    assert np.all(np.abs(t[0] - t[19]) <= 1e-6 * np.abs(t[19])), float(np.max(np.abs(t[0] - t[19]) / np.abs(t[19])))


def test_mse_histogram_route_one_key_workgroup_with_ragged_tail(ops):
    """workgroup 0's 65 536 whole-pack elements are ONE key (its 16-bit count carries) and the launch's n % 8 tail holds
    other keys: the single-key fallback must not credit the tail to that key"""
    dev = torch.device("cuda:0")
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    g = torch.Generator().manual_seed(77)
    # slab s of 16 Ki elements belongs to workgroup s % nwg (nwg = ceil(n / 65 536) = 65 here): slabs 0, nwg, 2 nwg, 3 nwg
    # -- workgroup 0's -- hold the constant
    n = 65536 * 64 + 5
    nwg = 65
    assert nwg <= cus
    x = (torch.randn(n, generator=g) * 0.5).bfloat16()
    for j in range(4):
        s0 = (j * nwg) * 16384
        x[s0:s0 + 16384] = 0.75
    x[-5:] = torch.tensor([7.0, -6.0, 5.0, -4.0, 3.0]).bfloat16()
    t = _tables(ops, x.to(dev), -8, 7, True)
    assert np.all(np.abs(t[0] - t[19]) <= 1e-6 * np.abs(t[19])), float(np.max(np.abs(t[0] - t[19]) / np.abs(t[19])))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_windowed_percentile_more_shards_than_one_call(ops, dtype):
    """70 cached batches (SBQ_MAX_BATCH = 64 per library call): the windowed protocol chunks them and adds the calls'
    records; (min, max) == the one-call engine on the concatenated data"""
    from sparsebit_amd import dist as sd
    from sparsebit_amd import select

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(123)
    shards = [(torch.randn(3001 + 17 * i, generator=g) * (1 + 0.05 * i)).to(dtype).to(dev) for i in range(70)]
    v = sd.run_lockstep([select.windowed_steps(shards, ops.HipWindowBackend(dtype), dev, percentile_alpha=1e-3)])[0]
    flat = torch.cat(shards)
    mn, mx = ops.percentile_select([flat], 1e-3, 0, False)
    assert float(v[0]) == float(mn) and float(v[1]) == float(mx)
    srt = torch.sort(flat.float())[0]
    neg, pos = int((flat.float() < 0).sum()), int((flat.float() >= 0).sum())
    n = flat.numel()
    assert float(v[1]) == float(srt[n - max(round(pos * 1e-3), 0) - 1]) and float(v[0]) == float(srt[max(round(neg * 1e-3), 1) - 1])


def test_unsupported_dtype_is_sbq_error(ops):
    from sparsebit_amd import lib as L

    with pytest.raises(L.SbqError):
        ops.HipWindowBackend(torch.float64)


def test_tuning_process_default_reaches_other_threads(ops):
    """knob 3 = 1 (never the resident schedule) set process-wide on this thread is what a fresh thread sees; a
    per-thread setting stays private.  Results are identical either way (the knob is an A/B switch): the check is that
    the call on the other thread runs and agrees."""
    from sparsebit_amd import lib as L

    dev = torch.device("cuda:0")
    x = torch.randn(512, 4096, device=dev).bfloat16()
    s = torch.rand(512, device=dev) * 0.05 + 0.01
    z = torch.zeros(512, device=dev)
    want = ops.fake_quant(x, s, z, -128, 127, 0)
    got = {}

    def other():
        torch.cuda.set_device(dev)
        got["y"] = ops.fake_quant(x, s, z, -128, 127, 0)
        torch.cuda.synchronize(dev)

    L.set_tuning(3, 1, process=True)
    try:
        th = threading.Thread(target=other)
        th.start()
        th.join()
    finally:
        L.set_tuning(3, 0, process=True)
    assert torch.equal(got["y"], want)
    with pytest.raises(L.SbqError):
        L.set_tuning(9, 0, process=True)


# --------------------------------------------------------------------------------------
# launch plans (sparsebit_amd.plan) and captured forwards (sparsebit_amd.graph)
# --------------------------------------------------------------------------------------
def _mk(scheme, bit, observer="MINMAX", target="weight", backend=None, quantizer="uniform", **kw):
    from sparsebit_amd.common import Backend
    from sparsebit_amd.config import quantizer_config
    from sparsebit_amd.quantizers import build_quantizer

    q = build_quantizer(quantizer_config(scheme, bit, quantizer=quantizer, observer=observer, target=target, **kw))
    q.set_backend(backend or Backend.VIRTUAL)
    return q


def _generic(q, x):
    """the same call with launch plans switched off: the definition of the result"""
    from sparsebit_amd import plan

    plan.set_enabled(False)
    try:
        with torch.no_grad():
            return q(x)
    finally:
        plan.set_enabled(True)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", ["w/channel", "a/tensor", "a/channel_nchw", "w/trt", "w/lsq"])
def test_planned_forward_equals_generic(case, dtype):
    from sparsebit_amd.common import Backend

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(hash(case) % 1000)
    if case == "w/channel":
        q, x = _mk("per-channel-symmetric", 8), torch.randn(96, 40, 3, 3, generator=g)
    elif case == "a/tensor":
        q, x = _mk("per-tensor-affine", 8, target="feature", layout="NCHW"), torch.randn(4, 24, 14, 14, generator=g)
    elif case == "a/channel_nchw":
        q, x = _mk("per-channel-affine", 8, target="feature", layout="NCHW"), torch.randn(4, 24, 14, 14, generator=g)
    elif case == "w/trt":
        q, x = _mk("per-channel-symmetric", 8, backend=Backend.TENSORRT), torch.randn(64, 129, generator=g)
    else:
        q, x = _mk("per-channel-symmetric", 4, quantizer="lsq"), torch.randn(64, 32, 3, 3, generator=g)
    x = x.to(dtype).to(dev)
    q = q.to(dev)
    q.update_observer(x)
    q.calc_qparams()
    q.enable_quant()
    want = _generic(q, x)
    with torch.no_grad():
        got = q(x)
        assert q._plans.plan is not None, "no plan was built"
        again = q(x)  # the cached plan
    assert got.dtype == want.dtype == torch.float32
    assert torch.equal(got, want) and torch.equal(again, want)
    # an own output dtype on THIS quantizer only (two models in one process may differ)
    q.keep_input_dtype = True
    other = _mk("per-channel-symmetric", 8).to(dev)
    with torch.no_grad():
        low = q(x)
    assert low.dtype == dtype and torch.equal(low, _generic(q, x)) and torch.equal(low.float(), want.to(dtype).float())
    assert other._out_keeps_dtype() is False
    # a non-contiguous input takes the generic route and still agrees
    xt = x.transpose(0, 1) if x.dim() == 2 else x.permute(0, 1, 3, 2)
    if case != "a/channel_nchw" and not case.startswith("w/"):
        with torch.no_grad():
            assert torch.equal(q(xt), _generic(q, xt))


def test_plan_follows_recalibration_and_switches():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(8)
    w1 = torch.randn(64, 256, generator=g).to(dev)
    w2 = (torch.randn(64, 256, generator=g) * 3).to(dev)
    q = _mk("per-channel-symmetric", 8).to(dev)
    q.update_observer(w1)
    q.calc_qparams()
    q.enable_quant()
    with torch.no_grad():
        y1 = q(w1)
        p1 = q._plans.plan
        assert p1 is not None and q(w1) is not None and q._plans.plan is p1  # reused
        # re-calibration re-binds scale / zero_point: the old plan must not serve the next forward
        q.update_observer(w2)
        q.calc_qparams()
        y2 = q(w2)
        assert q._plans.plan is not p1
        assert torch.equal(y2, _generic(q, w2)) and not torch.equal(y2, y1)
        # an in-place update of the values needs no new plan and is seen by the next launch
        p2 = q._plans.plan
        q.scale.mul_(2.0)
        y3 = q(w2)
        assert q._plans.plan is p2 and torch.equal(y3, _generic(q, w2)) and not torch.equal(y3, y2)
        # the integer range
        q.set_bit(4)
        y4 = q(w2)
        assert torch.equal(y4, _generic(q, w2)) and y4.unique().numel() <= 16 * 64
        # switches
        q.disable_quant()
        assert q(w2) is w2
        q.enable_quant()
        q.enable_export_onnx()
        ye = q(w2.contiguous())  # torch builtins (the tracer's route), never the plan
        q.disable_export_onnx()
        assert ye.shape == w2.shape
        assert torch.equal(q(w2), _generic(q, w2))
    # autograd keeps the STE route
    wp = torch.nn.Parameter(w2.clone())
    y = q(wp)
    assert y.requires_grad and torch.equal(y.detach(), _generic(q, w2))


def test_planned_forward_does_not_sync():
    from sparsebit_amd.common import Backend

    dev = torch.device("cuda:0")
    x = torch.randn(8, 16, 14, 14, device=dev)
    w = torch.randn(32, 16, 3, 3, device=dev)
    qa = _mk("per-tensor-symmetric", 8, target="feature", layout="NCHW", backend=Backend.TENSORRT).to(dev)
    qw = _mk("per-channel-symmetric", 8, backend=Backend.TENSORRT).to(dev)
    for q, t in ((qa, x), (qw, w)):
        q.update_observer(t)
        q.calc_qparams()
        q.enable_quant()
    with torch.no_grad():
        qa(x), qw(w)
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")
        try:
            for _ in range(200):
                qa(x), qw(w)
        finally:
            torch.cuda.set_sync_debug_mode("default")


def _resnet20():
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from examples import resnet20_quantopr as R

    return R


def test_captured_forward_equals_eager_and_follows_recalibration():
    """hipGraph of a whole quantized ResNet-20 forward: replay == eager bit for bit, new input values flow through the
    static buffer, a re-calibration invalidates the capture (re-captured by default, StaleCapture on request)"""
    from sparsebit_amd import graph as G
    from sparsebit_amd.calibration import DeviceCalibrator

    R = _resnet20()
    dev = torch.device("cuda:0")
    model, x = R.build(dev)
    with torch.no_grad():
        want = model(x).clone()
    fwd = G.capture(model, x)
    assert fwd.captures == 1 and not fwd.stale()
    assert torch.equal(fwd(x), want)
    g = torch.Generator().manual_seed(99)
    x2 = torch.randn(x.shape, generator=g).to(dev)
    with torch.no_grad():
        want2 = model(x2).clone()
    assert torch.equal(fwd(x2), want2) and fwd.captures == 1
    # in-place change of a step size: no re-capture, the replay reads the new value
    q0 = model.stem.weight_quantizer
    with torch.no_grad():
        q0.scale.mul_(1.5)
        want3 = model(x2).clone()
    assert torch.equal(fwd(x2), want3) and fwd.captures == 1 and not torch.equal(want3, want2)
    # re-calibration on other data re-binds every scale / zero_point
    strict = G.capture(model, x, on_stale="raise")
    data = [torch.randn(x.shape, generator=g).to(dev) * 2.0 for _ in range(2)]
    DeviceCalibrator(model).calibrate(data)
    for q in R.quantizers(model):
        q.enable_quant()
    assert fwd.stale()
    with pytest.raises(G.StaleCapture):
        strict(x2)
    with torch.no_grad():
        want4 = model(x2).clone()
    got4 = fwd(x2)
    assert fwd.captures == 2 and torch.equal(got4, want4) and not torch.equal(want4, want3)
    with pytest.raises(RuntimeError):
        G.capture(model.train(), x)
    model.eval()


def test_e2e_resnet20_plan_and_graph_are_identical_and_the_replay_is_faster():
    R = _resnet20()
    rec = R.measure(torch.device("cuda:0"), iters=100)
    assert rec["plan_equals_eager"] and rec["graph_equals_eager"], rec
    # (wall-clock ordering with a wide margin only: the replay was 2.4-2.6x faster than the generic route on every box so far)
    assert rec["graph_us"] < rec["eager_us"], rec


def test_captured_forward_with_frozen_weights():
    """freeze_weights=True: the weight quantizers run once before the capture, the replay equals the eager forward"""
    from sparsebit_amd import graph as G

    R = _resnet20()
    dev = torch.device("cuda:0")
    model, x = R.build(dev, seed=3)
    with torch.no_grad():
        want = model(x).clone()
    fwd = G.capture(model, x, freeze_weights=True)
    assert torch.equal(fwd(x), want)
    assert len(fwd._frozen) == 22 and all(q._pregrouped is None for q in R.quantizers(model))
    g = torch.Generator().manual_seed(5)
    x2 = torch.randn(x.shape, generator=g).to(dev)
    with torch.no_grad():
        want2 = model(x2).clone()
    assert torch.equal(fwd(x2), want2)


# --------------------------------------------------------------------------------------
# GPTQ 4-bit batched mat-mul on the fp32 matrix cores (csrc/sbq_gptq.hip: gptq_mfma_kernel, 5 <= B <= 32)
# --------------------------------------------------------------------------------------
def _gptq_case(in_f, out_f, gs, B, seed, dev):
    g = torch.Generator().manual_seed(seed)
    groups = in_f // gs if gs else 1
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (in_f // 8, out_f), generator=g, dtype=torch.int64).to(torch.int32)
    sc = (torch.rand(out_f, groups, generator=g) * 0.02 + 0.001).float()
    zr = (torch.randint(0, 16, (out_f, groups), generator=g).float() * sc).float()
    x = torch.randn(B, in_f, generator=g).float()
    bias = torch.randn(out_f, generator=g).float()
    return qw, sc, zr, x, bias


@pytest.mark.parametrize("batch", [5, 7, 8, 9, 15, 16, 17, 24, 29, 31, 32, 33, 64, 100])
@pytest.mark.parametrize("in_f,out_f,gs", [
    (128, 64, 128),      # one block, one tile: three of the workgroup's four waves have no K chunk
    (384, 128, 128),     # three blocks: an odd chunk (the dropped dead block)
    (640, 192, 128),     # five blocks over four waves
    (1024, 4096, 0),     # one group for the whole of K
    (2048, 1024, 256),   # a group spans two blocks
    (4096, 4096, 128),   # the decode shape: 2 blocks per wave, 4 K blocks of workgroups, the arrival fold
    (11008, 4096, 128),  # 86 blocks: uneven chunks, 6 K blocks
    (4096, 11008, 128),  # 172 tiles (not a multiple of 8: no XCD swizzle)
])
def test_gptq_batched_mfma_vs_oracle(ops, oracle_mod, batch, in_f, out_f, gs):
    """y == oracle(cuda_kernel_4bit.cu:36-180) at the reference test's rtol = atol = 1e-5 (test_cuda_kernel.py:81-126
    runs these batch sizes), bias kept, a second call gives the same bits (deterministic fold), and the strip tiles of
    four rows this kernel replaced (knob 2 = 26) agree to the same tolerance"""
    from sparsebit_amd import lib as L

    dev = torch.device("cuda:0")
    if batch not in (8, 17, 32, 100) and in_f * out_f > 4096 * 4096:
        pytest.skip("large shapes: three batch sizes")
    qw, sc, zr, x, bias = _gptq_case(in_f, out_f, gs, batch, 3 + in_f % 89 + batch, dev)
    qwd, scd, zrd, xd = qw.to(dev), sc.to(dev), zr.to(dev), x.to(dev)
    outs = {}
    for knob in (0, 26):
        L.set_tuning(2, knob)
        try:
            y = bias.repeat(batch, 1).to(dev)
            ops.vecquant4matmul(xd, qwd, y, scd, zrd, gs)
            y2 = bias.repeat(batch, 1).to(dev)
            ops.vecquant4matmul(xd, qwd, y2, scd, zrd, gs)
            assert torch.equal(y, y2), "knob %d: two calls differ" % knob
            outs[knob] = y.cpu().numpy()
        finally:
            L.set_tuning(2, 0)
    ref = oracle_mod.vecquantmatmul(x.numpy(), qw.numpy(), bias.numpy(), sc.numpy(), zr.numpy(), gs, 4)
    # an atol SCALED by max|y| (these are random shapes with |y| up to a few hundred, summed in a different fp32 order
    # than the oracle's; the reference's own shapes are held to its literal rtol = atol = 1e-5 in test_gpu_r02.py)
    tol = 1e-5 * max(1.0, float(np.abs(ref).max()))
    for knob, got in outs.items():
        assert np.all(np.abs(got - ref) <= tol + 1e-5 * np.abs(ref)), (knob, float(np.abs(got - ref).max()))


def test_gptq_batched_mfma_inf_nan_rows(ops):
    """an inf / NaN activation poisons its OWN batch row only (the dead block of an odd chunk is dropped by a select,
    not multiplied by zero; rows >= batch are clamped copies that are never stored)"""
    dev = torch.device("cuda:0")
    qw, sc, zr, x, bias = _gptq_case(384, 128, 128, 9, 41, dev)
    x[2, 300] = float("inf")
    x[5, 17] = float("nan")
    y = bias.repeat(9, 1).to(dev)
    ops.vecquant4matmul(x.to(dev), qw.to(dev), y, sc.to(dev), zr.to(dev), 128)
    y = y.cpu()
    bad = ~torch.isfinite(y).all(dim=1)
    assert bad.tolist() == [False, False, True, False, False, True, False, False, False]


def test_plan_cache_keeps_the_signatures_of_a_shared_quantizer():
    """ONE quantizer fed two shapes in turn (the reference's QAdd sends both addends through one input quantizer; a
    shared quantizer may also see two layouts): both plans stay, neither is rebuilt on the next round, results equal
    the generic route"""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(4)
    a = torch.randn(8, 16, 14, 14, generator=g).to(dev)
    b = torch.randn(8, 16, 7, 7, generator=g).to(dev)
    q = _mk("per-tensor-affine", 8, target="feature", layout="NCHW").to(dev)
    q.update_observer(a)
    q.calc_qparams()
    q.enable_quant()
    with torch.no_grad():
        ya, yb = q(a), q(b)
        pa_b = {id(q._plans.plan)} | {id(m) for m in q._plans.more}
        assert len(pa_b) == 2
        for _ in range(3):
            assert torch.equal(q(a), ya) and torch.equal(q(b), yb)
        assert {id(q._plans.plan)} | {id(m) for m in q._plans.more} == pa_b  # nothing rebuilt
        assert torch.equal(ya, _generic(q, a)) and torch.equal(yb, _generic(q, b))
        q.set_bit(4)  # a structural change drops the old plans instead of keeping them alive
        q(a)
        assert q._plans.more == []
