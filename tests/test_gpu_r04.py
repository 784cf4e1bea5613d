"""Round-4 changes on the device (all through the C ABI / ops.py / the product classes):
  * TensorRT-backend forwards do not synchronise the host (quant_tensor.py:128-156),
  * GPTQ's grid search on LONG rows (perchannel=False flattens the weight: quant.py:86-104) == the wave-per-row kernel
    and == the reference's tensor ops,
  * zero-contract workspaces: a dirty workspace is cleaned by the library, a busy one refused (include/sbq.h 5b),
  * the multi-matrix mat-vec with unequal widths, the grouped k-th value with more items than fit the
    chip at four slabs per workgroup, calibrate_forward on a disabled quantizer,
  * the QDQ-ONNX artifact from quantizers calibrated on the device: DequantizeLinear(file) == fake-quant output.
"""
import ctypes
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from sparsebit_amd import ops as _ops

    return _ops


def _mk(scheme, bit, observer="MINMAX", target="weight", backend=None, quantizer="uniform", **kw):
    from sparsebit_amd.common import Backend
    from sparsebit_amd.config import quantizer_config
    from sparsebit_amd.quantizers import build_quantizer

    q = build_quantizer(quantizer_config(scheme, bit, quantizer=quantizer, observer=observer, target=target, **kw))
    q.set_backend(backend or Backend.VIRTUAL)
    return q


def test_trt_forward_does_not_sync():
    """1000 TensorRT-backend forwards (weight per channel + activation per tensor) under
    torch.cuda.set_sync_debug_mode("error"): the reference's `assert abs(zero_point).sum() == 0` is a device-to-host
    read per forward; here it is made once per zero-point tensor (version)."""
    from sparsebit_amd.common import Backend

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    w = torch.randn(64, 32, 3, 3, generator=g).to(dev)
    a = torch.randn(8, 32, 14, 14, generator=g).to(dev)
    qw = _mk("per-channel-symmetric", 8, backend=Backend.TENSORRT).to(dev)
    qa = _mk("per-tensor-symmetric", 8, target="feature", backend=Backend.TENSORRT).to(dev)
    for q, x in ((qw, w), (qa, a)):
        q.update_observer(x)
        q.calc_qparams()
        q.enable_quant()
    with torch.no_grad():
        ref_w, ref_a = qw(w), qa(a)  # first forwards: the one host read per zero-point tensor happens here
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")
        try:
            for _ in range(1000):
                yw, ya = qw(w), qa(a)
        finally:
            torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    assert torch.equal(yw, ref_w) and torch.equal(ya, ref_a)
    # the assertion itself is intact: an asymmetric zero point is still refused -- also after an in-place change of
    # a tensor that had been validated
    qa.zero_point.add_(3.0)
    with pytest.raises(AssertionError, match="tensorrt only support symmetric quant"):
        qa(a)
    qa.zero_point.zero_()
    with torch.no_grad():
        assert torch.equal(qa(a), ref_a)


def _ref_find_params_mse(x2d, maxq, sym, norm=2.4, grid=100, maxshrink=0.8):
    """quant.py:71-104 on torch ops (the reference's own arithmetic), rows of x2d"""
    xmin = torch.minimum(x2d.min(1)[0], torch.zeros(x2d.shape[0], device=x2d.device))
    xmax = torch.maximum(x2d.max(1)[0], torch.zeros(x2d.shape[0], device=x2d.device))
    if sym:
        xmax = torch.maximum(torch.abs(xmin), xmax)
        xmin = torch.where(xmin < 0, -xmax, xmin)
    scale = (xmax - xmin) / maxq
    zero = torch.full_like(scale, (maxq + 1) / 2) if sym else torch.round(-xmin / scale)
    best = torch.full([x2d.shape[0]], float("inf"), device=x2d.device)
    errs = []
    for i in range(int(maxshrink * grid)):
        p = 1 - i / grid
        xmin1, xmax1 = p * xmin, p * xmax
        scale1 = (xmax1 - xmin1) / maxq
        zero1 = torch.round(-xmin1 / scale1) if not sym else zero
        q = torch.clamp(torch.round(x2d / scale1.unsqueeze(1)) + zero1.unsqueeze(1), 0, maxq)
        err = torch.sum((scale1.unsqueeze(1) * (q - zero1.unsqueeze(1)) - x2d).abs().pow(norm), 1)
        errs.append(err)
        tmp = err < best
        best = torch.where(tmp, err, best)
        scale = torch.where(tmp, scale1, scale)
        zero = torch.where(tmp, zero1, zero)
    return scale, zero, torch.stack(errs, 1)


@pytest.mark.parametrize("sym", [True, False])
@pytest.mark.parametrize("bit", [4, 3])
def test_gptq_mse_search_long_rows(ops, sym, bit):
    """find_params(perchannel=False, mse=True): one row of 1.05 M elements (sliced across workgroups, fp64 partials)
    against the reference's tensor ops; a differing candidate is accepted only where the reference's own fp32 error
    sums of the two candidates tie to 1e-6 relative (torch.sum's order is not ours).  Rows just above and below the
    split threshold agree with each other on the same data."""
    from sparsebit_amd import gptq

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(40 + bit)
    w = (torch.randn(1024, 1027, generator=g) * 0.02).to(dev)
    q = gptq.Quantizer()
    q.configure(bit, perchannel=False, sym=sym, mse=True)
    q.find_params(w, weight=True)
    # (the reference arithmetic on the HOST, like the reference's CPU path: torch's GPU `tensor / python_scalar` is a
    # reciprocal multiply, one ulp off the correctly rounded quotient the kernels -- and torch's CPU ops -- produce)
    s_ref, z_ref, errs = _ref_find_params_mse(w.cpu().reshape(1, -1), 2 ** bit - 1, sym)
    s_got, z_got = q.scale.reshape(-1)[0].cpu(), q.zero.reshape(-1)[0].cpu()
    assert q.scale.shape == (1024, 1) and bool((q.scale == s_got).all())
    if not (torch.equal(s_got, s_ref[0]) and torch.equal(z_got, z_ref[0])):
        # which candidates? scale1 = p * range / maxq identifies p
        e = errs[0]
        i_ref = int(torch.argmin(e))
        cand = [(abs(float(s_got) - float(s_ref[0] * (1 - i / 100) / (1 - i_ref / 100))), i) for i in range(80)]
        i_got = min(cand)[1]
        assert abs(float(e[i_got]) - float(e[i_ref])) <= 1e-6 * float(e[i_ref]), (i_got, i_ref, float(e[i_got]), float(e[i_ref]))
    # split path (inner > 16384) vs wave-per-row path on the same rows: 3 rows of 20000 vs the same data as 6 rows of 10000
    x = (torch.randn(3, 20000, generator=g) * 0.05).to(dev)
    maxq = 2 ** bit - 1
    s1, z1, e1 = _ref_find_params_mse(x.cpu(), maxq, sym)
    xmin = torch.minimum(x.min(1)[0], torch.zeros(3, device=dev))
    xmax = torch.maximum(x.max(1)[0], torch.zeros(3, device=dev))
    if sym:
        xmax = torch.maximum(torch.abs(xmin), xmax)
        xmin = torch.where(xmin < 0, -xmax, xmin)
    scale = ((xmax - xmin) / maxq).contiguous()
    zero = (torch.full_like(scale, (maxq + 1) / 2) if sym else torch.round(-xmin / scale)).contiguous()
    idx = ops.gptq_mse_search(x, xmin.contiguous(), xmax.contiguous(), maxq, sym, scale, zero)
    for r in range(3):
        i_ref = int(torch.argmin(e1[r]))
        i_got = int(idx[r])
        assert i_got == i_ref or abs(float(e1[r, i_got]) - float(e1[r, i_ref])) <= 1e-6 * float(e1[r, i_ref])
        if i_got == i_ref:
            assert float(scale[r]) == float(s1[r]) and float(zero[r]) == float(z1[r])


def test_dirty_workspace_is_cleaned_and_busy_workspace_refused(ops):
    """include/sbq.h 5b: a selection workspace full of garbage gives the exact rank (the library zeroes a region it
    has not seen); the same workspace used from a second stream while the first is busy is refused with SBQ_ERR_BUSY;
    after release (or once the first stream is idle) it is accepted again."""
    from sparsebit_amd import lib as L

    lib = L.load()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1 << 22, generator=g).bfloat16().to(dev)
    k = x.numel() // 3
    want = float(torch.sort(x.float().abs())[0][k - 1])
    nbytes = lib.sbq_radix_select_workspace_bytes(1, 1)
    ws = torch.full((nbytes,), 0xA5, dtype=torch.uint8, device=dev)  # dirty on purpose
    lib.sbq_workspace_release(L.ptr(ws), ws.numel())  # (the allocator may hand out an address the library has seen)
    out = torch.empty((), dtype=torch.float32, device=dev)
    s1 = torch.cuda.current_stream(dev)
    rc = lib.sbq_kth_value(L.ptr(x), L.BF16, x.numel(), 1, k, L.ptr(out), L.ptr(ws), ws.numel(), ctypes.c_void_p(s1.cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    assert float(out) == want
    # busy: keep stream 1 occupied, call on stream 2 with the same workspace
    s2 = torch.cuda.Stream(device=dev)
    big = torch.randn(1 << 26, device=dev)
    for _ in range(20):
        big = big * 1.0001 + 1.0
    rc = lib.sbq_kth_value(L.ptr(x), L.BF16, x.numel(), 1, k, L.ptr(out), L.ptr(ws), ws.numel(), ctypes.c_void_p(s1.cuda_stream))
    assert rc == 0
    rc2 = lib.sbq_kth_value(L.ptr(x), L.BF16, x.numel(), 1, k, L.ptr(out), L.ptr(ws), ws.numel(), ctypes.c_void_p(s2.cuda_stream))
    assert rc2 == 8, rc2  # SBQ_ERR_BUSY
    with pytest.raises(L.SbqError, match="still in use"):
        L.check(rc2)
    torch.cuda.synchronize()
    rc3 = lib.sbq_kth_value(L.ptr(x), L.BF16, x.numel(), 1, k, L.ptr(out), L.ptr(ws), ws.numel(), ctypes.c_void_p(s2.cuda_stream))
    assert rc3 == 0  # stream 1 idle: the workspace moves to stream 2
    torch.cuda.synchronize()
    assert float(out) == want
    # release + scribble + reuse: cleaned again
    assert lib.sbq_workspace_release(L.ptr(ws), ws.numel()) == 0
    ws.fill_(0x5A)
    torch.cuda.synchronize()
    rc4 = lib.sbq_kth_value(L.ptr(x), L.BF16, x.numel(), 1, k, L.ptr(out), L.ptr(ws), ws.numel(), ctypes.c_void_p(s1.cuda_stream))
    assert rc4 == 0
    torch.cuda.synchronize()
    assert float(out) == want
    # the GPTQ mat-vec's counters: dirty workspace, right answer
    from oracle import oracle as O

    in_f, out_f = 1024, 256
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (in_f // 8, out_f), generator=g, dtype=torch.int64).to(torch.int32).to(dev)
    sc = (torch.rand(out_f, in_f // 128, generator=g) * 0.02 + 0.001).to(dev)
    zr = (torch.randint(0, 16, (out_f, in_f // 128), generator=g).float().to(dev) * sc)
    xv = torch.randn(1, in_f, generator=g).to(dev)
    y = torch.zeros(1, out_f, device=dev)
    gws = torch.full((lib.sbq_gptq_workspace_bytes(1, in_f, out_f),), 0xEE, dtype=torch.uint8, device=dev)
    lib.sbq_workspace_release(L.ptr(gws), gws.numel())
    rc = lib.sbq_vecquant4matmul(L.ptr(xv), L.ptr(qw), L.ptr(y), L.ptr(sc), L.ptr(zr), 1, in_f, out_f, 128, L.ptr(gws), gws.numel(),
                                 ctypes.c_void_p(s1.cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    ref = O.vecquantmatmul(xv.cpu().numpy(), qw.cpu().numpy(), np.zeros(out_f, np.float32), sc.cpu().numpy(), zr.cpu().numpy(), 128, 4)
    assert np.allclose(y.cpu().numpy(), ref, rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(ref).max())))


def test_multi_matvec_unequal_widths_workspace(ops):
    """out = [1024, 257] (257 is not a multiple of 32: the per-matrix route) with a deep in_features: the
    workspace is sized by sbq_vecquantmatmul_multi_workspace_bytes, and each output equals its single call"""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    in_f = 28672
    outs_f = [1024, 257]
    x = torch.randn(1, in_f, generator=g).to(dev)
    mats = []
    for o in outs_f:
        qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (in_f // 8, o), generator=g, dtype=torch.int64).to(torch.int32).to(dev)
        sc = (torch.rand(o, in_f // 128, generator=g) * 0.02 + 0.001).to(dev)
        zr = torch.randint(0, 16, (o, in_f // 128), generator=g).float().to(dev) * sc
        mats.append((qw, sc, zr))
    ys = [torch.zeros(1, o, device=dev) for o in outs_f]
    ops.vecquantmatmul_multi(4, x, [m[0] for m in mats], ys, [m[1] for m in mats], [m[2] for m in mats], 128)
    for (qw, sc, zr), y, o in zip(mats, ys, outs_f):
        one = torch.zeros(1, o, device=dev)
        ops.vecquantmatmul(4, x, qw, one, sc, zr, 128)
        assert torch.allclose(y, one, rtol=1e-5, atol=1e-4)


def test_group_kth_more_wishes_than_compute_units(ops):
    """64 tensors of 1.2 M elements each want 19 workgroups apiece (1216 for a 256-CU chip): the launch is scaled to
    one sitting and every threshold still equals the per-tensor call"""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(12)
    for dt in (torch.bfloat16, torch.float32):
        ts = [(torch.randn(1200 * 1000 + 8 * i, generator=g) * (1 + i % 5)).to(dt).to(dev) for i in range(64)]
        ks = [1 + (t.numel() * (i % 7 + 1)) // 9 for i, t in enumerate(ts)]
        got = ops.group_kth_value(ts, ks, True)
        ref = torch.stack([ops.kth_value(t, k, True) for t, k in zip(ts, ks)])
        assert torch.equal(got, ref)
        j = 17
        assert float(got[j]) == float(torch.sort(ts[j].float().abs())[0][ks[j] - 1])


def test_calibrate_forward_on_a_disabled_quantizer_twice():
    """QUANTIZER.DISABLE -> calibrate_forward is the identity, nothing is cached, a second call is fine"""
    dev = torch.device("cuda:0")
    q = _mk("per-channel-symmetric", 8, disable=True).to(dev)
    w = torch.randn(16, 64, device=dev)
    for _ in range(2):
        y = q.calibrate_forward(w)
        assert torch.equal(y, w) and len(q.observer.data_cache) == 0


def test_qdq_onnx_file_from_device_quantizers(tmp_path, ops):
    """export.save_qdq_onnx on a QuantOpr-style model calibrated on the device: the file parses back (own reader; the
    CPU test checks the bytes with google.protobuf) and DequantizeLinear of the stored levels equals the fake-quant
    forward bit for bit (quant_model.py:222-324: the constants of the reference's QDQ graph, `bits` included)"""
    from sparsebit_amd import export

    dev = torch.device("cuda:0")

    class Op(torch.nn.Module):
        def __init__(self, mod, wq, aq):
            super().__init__()
            self.fwd, self.weight = mod, mod.weight
            self.weight_quantizer, self.input_quantizer = wq, aq

    torch.manual_seed(3)
    net = torch.nn.Module()
    net.conv = Op(torch.nn.Conv2d(8, 16, 3), _mk("per-channel-symmetric", 8), _mk("per-tensor-affine", 8, target="feature"))
    net.fc = Op(torch.nn.Linear(64, 10), _mk("per-channel-symmetric", 4, quantizer="lsq"), _mk("per-tensor-symmetric", 4, target="feature"))
    net = net.to(dev)
    g = torch.Generator().manual_seed(4)
    for op, a in ((net.conv, torch.randn(4, 8, 12, 12, generator=g)), (net.fc, torch.randn(4, 64, generator=g))):
        for q, x in ((op.weight_quantizer, op.weight.detach()), (op.input_quantizer, a.to(dev))):
            q.update_observer(x)
            q.calc_qparams()
            q.enable_quant()
    path = os.path.join(str(tmp_path), "net_qdq.onnx")
    n = export.save_qdq_onnx(net, path)
    assert n == os.path.getsize(path)
    back = export.load_qdq_onnx(path)
    assert sorted(back["weights"]) == ["conv.weight", "fc.weight"] and sorted(back["activations"]) == ["conv.input", "fc.input"]
    for name, op in (("conv", net.conv), ("fc", net.fc)):
        rec = back["weights"][name + ".weight"].to(dev)
        with torch.no_grad():
            want = op.weight_quantizer(op.weight.detach())
        assert rec.bits == op.weight_quantizer.bit and rec.axis == 0 and rec.signed
        assert torch.equal(rec.dequantize(), want.float())
        a = back["activations"][name + ".input"]
        iq = op.input_quantizer
        assert a["bits"] == iq.bit and a["axis"] is None
        assert float(a["scale"]) == float(iq.scale.reshape(-1)[0]) and int(a["zero_point"]) == int(iq.zero_point.round().reshape(-1)[0])


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_streaming_minmax_state_equals_torch(ops, dtype):
    """sbq_minmax_accumulate over batches of many sizes (ragged tails, one element, exact chunk multiples) == torch's
    min / max of the union; NaN anywhere pins both; the signs of zero are values, not bit patterns"""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(21)
    sizes = [1, 7, 8, 4096, 4097, 16384 + 24, 3 * 4096 * 4, 1_000_003]
    st = ops.minmax_state(dev)
    lo = hi = None
    for i, n in enumerate(sizes):
        x = (torch.randn(n + 8, generator=g) * (1 + i)).to(dtype).to(dev)[:n]  # (a 16-byte aligned view of n elements)
        assert ops.minmax_accumulate(x, st)
        lo = x.float().min() if lo is None else torch.minimum(lo, x.float().min())
        hi = x.float().max() if hi is None else torch.maximum(hi, x.float().max())
        a, b = ops.minmax_state_read(st)
        assert float(a) == float(lo) and float(b) == float(hi), (n, float(a), float(lo), float(b), float(hi))
    # all negative / all positive batches into fresh states
    for sign in (-1.0, 1.0):
        st2 = ops.minmax_state(dev)
        x = (sign * (torch.rand(70001, generator=g) + 0.5)).to(dtype).to(dev)
        ops.minmax_accumulate(x, st2)
        a, b = ops.minmax_state_read(st2)
        assert float(a) == float(x.float().min()) and float(b) == float(x.float().max())
    # infinities are values; NaN wins
    x = torch.tensor([1.0, float("inf"), -2.0, float("-inf")] * 4, dtype=dtype, device=dev)
    st3 = ops.minmax_state(dev)
    ops.minmax_accumulate(x, st3)
    a, b = ops.minmax_state_read(st3)
    assert float(a) == float("-inf") and float(b) == float("inf")
    y = torch.zeros(9000, dtype=dtype, device=dev)
    y[8191] = float("nan")
    ops.minmax_accumulate(y, st3)
    a, b = ops.minmax_state_read(st3)
    assert torch.isnan(a).all() and torch.isnan(b).all()
    # not eligible: an unaligned view -> the caller falls back
    assert ops.minmax_accumulate(torch.zeros(100, dtype=dtype, device=dev)[1:], ops.minmax_state(dev)) is False


def test_streaming_observer_equals_cached_observer():
    """observers/minmax.py consume() (one launch per batch) == the cache-then-reduce protocol, per tensor, including a
    mix of streamed and cached batches and an unaligned batch in between"""
    from sparsebit_amd.common import Backend

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(8)
    batches = [(torch.randn(4, 16, 14, 14, generator=g) * (1 + 0.5 * i)).to(dev) for i in range(4)]
    odd = torch.randn(1000 + 1, generator=g).to(dev)[1:]  # 4-byte aligned only
    qa = _mk("per-tensor-affine", 8, target="feature").to(dev)
    qb = _mk("per-tensor-affine", 8, target="feature").to(dev)
    qa.dims = 4
    for b in batches[:3]:
        qa.observer.consume(b)
    qa.observer.consume(odd)
    qa.update_observer(batches[3])  # a cached one joins
    for b in batches:
        qb.update_observer(b)
    qb.update_observer(odd)
    sa, za = qa.calc_qparams()
    sb, zb = qb.calc_qparams()
    assert torch.equal(sa.reshape(-1), sb.reshape(-1)) and torch.equal(za.reshape(-1), zb.reshape(-1))
    assert torch.equal(qa.observer.min_val.reshape(-1), qb.observer.min_val.reshape(-1))
    assert qa.observer._state is None and qa.observer._running is None  # back to 'nothing seen'


# --------------------------------------------------------------------------------------
# the full-histogram selection engine (h16_select_kernel): 16-bit tensors of up to 65 536 elements per compute unit
# --------------------------------------------------------------------------------------
def _kth_ref(x, k, use_abs):
    v = x.float().abs() if use_abs else x.float()
    return float(torch.sort(v.reshape(-1))[0][k - 1])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_h16_engine_sizes_and_ranks(ops, dtype):
    """every size class the engine takes -- fewer elements than a pack would hold twice, ragged tails (n % 8 != 0),
    one workgroup, exactly one sitting of the chip (65 536 x CUs) -- at extreme, tail and bulk ranks, against torch.sort;
    one element more than a sitting goes to win_one_kernel and agrees as well"""
    dev = torch.device("cuda:0")
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    g = torch.Generator().manual_seed(77)
    for n in (8, 9, 15, 100, 4099, 65536, 65536 + 13, 300001, 65536 * cus, 65536 * cus + 8):
        x = (torch.randn(n, generator=g) * torch.logspace(-2, 1, n)[torch.randperm(n, generator=g)]).to(dtype).to(dev)
        for use_abs in (False, True):
            for k in sorted({1, 2, max(1, n // 1000), n // 3 + 1, n // 2 + 1, n - 1, n}):
                got = float(ops.kth_value(x, k, use_abs))
                assert got == _kth_ref(x, k, use_abs), (n, k, use_abs, got)
        for alpha in (1e-3, 0.1, 1e-6):
            mn, mx = ops.percentile_select([x], alpha, 0, False)
            xf = x.float()
            neg, pos = int((xf < 0).sum()), int((xf >= 0).sum())
            want_mx = _kth_ref(x, n - max(round(pos * alpha), 0), False) if pos > 0 else 0.0
            want_mn = _kth_ref(x, max(round(neg * alpha), 1), False) if neg > 0 else 0.0
            assert float(mn) == want_mn and float(mx) == want_mx, (n, alpha, float(mn), want_mn, float(mx), want_mx)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_h16_engine_adversarial_data(ops, dtype):
    """what breaks histograms and samples: one repeated value filling whole workgroups (a 16-bit count of 65 536
    carries into its neighbour), zeros everywhere (ReLU), two values, sorted data, data periodic with the sample's
    stride, NaN / inf mixed in, a single outlier the sample cannot see"""
    dev = torch.device("cuda:0")
    n = 65536 * 8 + 24
    g = torch.Generator().manual_seed(5)
    cases = {
        "constant": torch.full((n,), 0.375),
        "constant_neg": torch.full((n,), -3.0),
        "all_zero": torch.zeros(n),
        "neg_zero": torch.full((n,), -0.0),
        "relu": torch.relu(torch.randn(n, generator=g)),
        "two_values": torch.where(torch.rand(n, generator=g) < 0.3, torch.tensor(1.5), torch.tensor(-0.25)),
        "sorted": torch.sort(torch.randn(n, generator=g))[0],
        "periodic": torch.arange(n).remainder(2048).float() / 64.0 - 10.0,
        "outlier": torch.cat([torch.randn(n - 1, generator=g) * 0.01, torch.tensor([1000.0])]),
        "specials": torch.cat([torch.randn(n - 6, generator=g), torch.tensor([float("inf"), float("-inf"), float("nan"), float("nan"), 0.0, -0.0])]),
        "one_block_constant": torch.cat([torch.full((65536 * 2,), 7.0), torch.randn(n - 65536 * 2, generator=g)]),
    }
    for name, t in cases.items():
        x = t.to(dtype).to(dev)
        xf = x.float()
        srt = torch.sort(xf.reshape(-1).cpu())[0].to(dev)  # on the host: NaN last, like kthvalue
        for k in (1, 7, n // 4, n // 2 + 1, n - 7, n):
            got = ops.kth_value(x, k, False)
            want = srt[k - 1]
            assert (torch.isnan(got) and torch.isnan(want)) or float(got) == float(want), (name, k, float(got), float(want))
        a = torch.sort(xf.abs().reshape(-1).cpu())[0].to(dev)
        for k in (1, n // 2 + 1, n):
            got = ops.kth_value(x, k, True)
            assert (torch.isnan(got) and torch.isnan(a[k - 1])) or float(got) == float(a[k - 1]), (name, k)
        if not torch.isnan(xf).any():
            mn, mx = ops.percentile_select([x], 1e-3, 0, False)
            neg, pos = int((xf < 0).sum()), int((xf >= 0).sum())
            want_mx = float(srt[n - max(round(pos * 1e-3), 0) - 1]) if pos > 0 else 0.0
            want_mn = float(srt[max(round(neg * 1e-3), 1) - 1]) if neg > 0 else 0.0
            assert float(mn) == want_mn and float(mx) == want_mx, (name, float(mn), want_mn, float(mx), want_mx)


def test_h16_engine_equals_win_one_engine(ops):
    """knob 2 = 18 selects round 3's one-launch engine: same values on the headline tensor, k-th value and percentile"""
    from sparsebit_amd import lib as L

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    w = (torch.randn(4096, 4096, generator=g) * torch.logspace(-2, 1, 4096).unsqueeze(1)).bfloat16().to(dev)
    res = {}
    for knob in (0, 18):
        L.set_tuning(2, knob)
        try:
            res[knob] = (float(ops.kth_value(w, w.numel() // 2 + 1, True)),) + tuple(float(v) for v in ops.percentile_select([w], 1e-3, 0, False))
        finally:
            L.set_tuning(2, 0)
    assert res[0] == res[18], res


# --------------------------------------------------------------------------------------
# the MSE observer of a 16-bit tensor taken as a whole: the histogram route (observers/mse.py:46-61)
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("scheme", ["per-tensor-symmetric", "per-tensor-affine"])
def test_mse_per_tensor_histogram_route(ops, oracle, dtype, scheme):
    """5.2 M elements (two launches' worth of workgroups would be 16.7 M; a ragged tail here), Gaussian with outliers
    and, in the second data set, half zeros: the 80 squared-error sums of the histogram route agree with the
    per-element route (knob 2 = 19) to 1e-6 relative and with the oracle's fp64 sums to 1e-5, and the chosen candidate is
    the oracle's (a differing index only where the oracle's own sums of the two candidates tie to 1e-7)"""
    from sparsebit_amd import lib as L

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(31)
    n = 5 * 1024 * 1024 + 4099
    sym = scheme.endswith("symmetric")
    qmin, qmax = (-128, 127) if sym else (0, 255)
    for relu in (False, True):
        x = torch.randn(n, generator=g) * 0.7
        x[::9973] *= 25.0
        if relu:
            x = torch.relu(x)
        x = x.to(dtype)
        xd = x.to(dev)
        mn, mx, _ = ops.channel_stats(xd, 0, False)
        tables = {}
        for knob in (0, 19):
            L.set_tuning(2, knob)
            try:
                sse = torch.zeros(1, L.MSE_CANDIDATES, dtype=torch.float64, device=dev)
                ops.mse_accumulate(xd, mn, mx, qmin, qmax, sym, sse, 0, False)
                tables[knob] = sse.cpu().numpy()[0]
            finally:
                L.set_tuning(2, 0)
        a, b = tables[0], tables[19]
        assert np.all(np.abs(a - b) <= 1e-6 * np.abs(b)), float(np.max(np.abs(a - b) / np.abs(b)))
        _, _, b_ref, sse_ref = oracle.mse(x.float().numpy().reshape(1, -1), qmin, qmax, sym, 0, False)
        assert np.all(np.abs(a - sse_ref[0]) <= 1e-5 * np.abs(sse_ref[0]))
        s, z, best = ops.mse_select(torch.from_numpy(a).reshape(1, -1).to(dev), n, mn, mx, qmin, qmax, sym)
        i_got, i_ref = int(best[0]), int(b_ref[0])
        assert i_got == i_ref or abs(sse_ref[0, i_got] - sse_ref[0, i_ref]) <= 1e-7 * sse_ref[0, i_ref], (i_got, i_ref)
    # a second call ADDS (two cached batches): twice the table
    sse2 = torch.zeros(1, L.MSE_CANDIDATES, dtype=torch.float64, device=dev)
    ops.mse_accumulate(xd, mn, mx, qmin, qmax, sym, sse2, 0, False)
    ops.mse_accumulate(xd, mn, mx, qmin, qmax, sym, sse2, 0, False)
    assert np.allclose(sse2.cpu().numpy()[0], 2 * tables[0], rtol=1e-12)


def test_mse_histogram_route_constant_block(ops):
    """65 536 x 64 equal elements (every workgroup's 16-bit count carries out of its half-dword) + a tail of other values"""
    from sparsebit_amd import lib as L

    dev = torch.device("cuda:0")
    x = torch.cat([torch.full((65536 * 64,), 0.5), torch.linspace(-1, 1, 70000)]).bfloat16().to(dev)
    mn, mx, _ = ops.channel_stats(x, 0, False)
    out = {}
    for knob in (0, 19):
        L.set_tuning(2, knob)
        try:
            sse = torch.zeros(1, L.MSE_CANDIDATES, dtype=torch.float64, device=dev)
            ops.mse_accumulate(x, mn, mx, -8, 7, True, sse, 0, False)
            out[knob] = sse.cpu().numpy()[0]
        finally:
            L.set_tuning(2, 0)
    assert np.all(np.abs(out[0] - out[19]) <= 1e-6 * np.abs(out[19]))


# --------------------------------------------------------------------------------------
# GPTQ mat-vec: the branch-free (LEAN) and the 512-thread single-pass (wide) strip kernels
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("bits", [4, 3, 2])
@pytest.mark.parametrize("batch", [1, 2])
@pytest.mark.parametrize("in_f,out_f,gs", [
    (4096, 4096, 128),    # one wide pass, every K lane live
    (2048, 1024, 128),    # half of the wide workgroup's K lanes are dead (clamped rows, zero activations)
    (4224, 2048, 128),    # one full wide pass + a 128-channel stub of a second
    (11008, 4096, 128),   # three wide passes, the last one partial; K split in the 256-thread route
    (4096, 11008, 0),     # un-grouped (one scale per column)
    (1152, 96, 128),      # 96 columns: three strips, no XCD swizzle
    (4096, 4096, 1024),   # groups wider than a K lane's 64 channels
])
def test_gptq_lean_and_wide_kernels(ops, bits, batch, in_f, out_f, gs):
    """the default route (LEAN loads; wide workgroups where the rule of gptq_matmul picks them), the 256-thread LEAN
    kernels (knob 2 = 25) and the branchy kernels of round 3 (knob 2 = 23) against the oracle -- tolerance 1e-5 of the
    largest output: the reference adds with atomicAdd, the order of the fp32 sums is not part of its contract
    (cuda_kernel_4bit.cu:64-81) -- and the same launch twice gives the same bits"""
    from oracle import oracle as O
    from sparsebit_amd import lib as L

    if gs and in_f % gs:
        pytest.skip("group size must divide in_features")
    g = torch.Generator().manual_seed(bits * 1000 + batch * 100 + in_f % 97)
    rows = (in_f + 31) // 32 * 3 if bits == 3 else (in_f * bits + 31) // 32
    groups = in_f // gs if gs else 1
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (rows, out_f), generator=g, dtype=torch.int64).to(torch.int32)
    sc = torch.rand(out_f, groups, generator=g) * 0.02 + 0.001
    zr = torch.randint(0, 2 ** bits, (out_f, groups), generator=g).float() * sc
    x = torch.randn(batch, in_f, generator=g)
    bias = torch.randn(out_f, generator=g)
    want = O.vecquantmatmul(x.numpy(), qw.numpy(), bias.numpy(), sc.numpy(), zr.numpy(), gs, bits)
    tol = 1e-5 * max(1.0, float(np.abs(want).max())) * max(1.0, (in_f / 4096) ** 0.5)
    dev = torch.device("cuda:0")
    qwd, scd, zrd, xd = qw.to(dev), sc.to(dev), zr.to(dev), x.to(dev)
    got = {}
    try:
        for knob in (0, 25, 23):
            L.set_tuning(2, knob)
            y = bias.repeat(batch, 1).contiguous().to(dev)
            ops.vecquantmatmul(bits, xd, qwd, y, scd, zrd, gs)
            y2 = bias.repeat(batch, 1).contiguous().to(dev)
            ops.vecquantmatmul(bits, xd, qwd, y2, scd, zrd, gs)
            assert torch.equal(y, y2), "knob %d: two launches differ" % knob
            got[knob] = y.cpu().numpy()
            assert np.abs(got[knob] - want).max() <= tol, "knob %d" % knob
    finally:
        L.set_tuning(2, 0)


def test_gptq_unaligned_activations_take_the_branchy_kernels(ops):
    """x 4 bytes off a 16-byte boundary: the LEAN kernels' 16-byte activation loads do not apply; same result"""
    from oracle import oracle as O

    g = torch.Generator().manual_seed(77)
    in_f, out_f = 4096, 2048
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (in_f // 8, out_f), generator=g, dtype=torch.int64).to(torch.int32)
    sc = torch.rand(out_f, in_f // 128, generator=g) * 0.02 + 0.001
    zr = torch.randint(0, 16, (out_f, in_f // 128), generator=g).float() * sc
    x = torch.randn(1, in_f, generator=g)
    want = O.vecquantmatmul(x.numpy(), qw.numpy(), np.zeros(out_f, np.float32), sc.numpy(), zr.numpy(), 128, 4)
    dev = torch.device("cuda:0")
    buf = torch.zeros(in_f + 8, device=dev)
    xd = buf[1:1 + in_f].view(1, in_f)
    xd.copy_(x)
    assert xd.data_ptr() % 16 == 4 and xd.is_contiguous()
    y = torch.zeros(1, out_f, device=dev)
    ops.vecquantmatmul(4, xd, qw.to(dev), y, sc.to(dev), zr.to(dev), 128)
    assert np.abs(y.cpu().numpy() - want).max() <= 1e-5 * max(1.0, float(np.abs(want).max()))
