"""Seeded differential fuzzing of the HIP path against the CPU oracle: random geometries
(outer / C / inner incl. primes, 1, pack-ragged), channel axes, dtypes, integer ranges, special
values sprinkled in, and storage offsets that break 16-byte alignment.  Every case is small (the
oracle finishes in milliseconds); the point is breadth over the dispatch logic -- ROWS / FLAT /
channels-last / scalar forward paths, vector vs element-wise reductions, grouped launches."""
import numpy as np
import pytest
import torch

from helpers import dev_tensor, same_values

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16, torch.float16]
RANGES = [(-128, 127), (0, 255), (-8, 7), (0, 15), (-2, 1), (-32768, 32767), (0, 3)]


def _shape(rng):
    nd = int(rng.integers(1, 5))
    pools = [1, 2, 3, 5, 7, 8, 16, 24, 31, 64, 100, 129, 256, 520]
    shape = [int(rng.choice(pools)) for _ in range(nd)]
    while int(np.prod(shape)) > 300000:
        shape[int(np.argmax(shape))] //= 2
    return tuple(max(s, 1) for s in shape)


def _data(rng, shape, dtype, offset):
    n = int(np.prod(shape))
    base = torch.from_numpy(rng.standard_normal(n + offset).astype(np.float32)) * float(rng.choice([0.05, 1.0, 30.0]))
    k = max(1, n // 50)
    idx = torch.from_numpy(rng.integers(0, n + offset, size=k))
    specials = torch.tensor([0.0, -0.0, 0.5, -0.5, 1.5, 2.5, float("inf"), -float("inf"), 1e-40, -1e-40, 3e38, float("nan")])
    base[idx] = specials[torch.from_numpy(rng.integers(0, len(specials), size=k))]
    storage = base.to(dtype).cuda()
    return storage[offset:].view(shape)  # contiguous, but its pointer is offset by `offset` elements


@pytest.mark.parametrize("seed", range(60))
def test_fuzz_forward_and_stats(oracle, seed):
    from sparsebit_amd import ops

    rng = np.random.default_rng(1000 + seed)
    shape = _shape(rng)
    dtype = DTYPES[seed % 3]
    offset = int(rng.choice([0, 0, 1, 3, 4, 8]))
    x = _data(rng, shape, dtype, offset)
    xf = x.float().cpu().numpy()
    per_channel = bool(rng.integers(0, 2))
    ch_axis = int(rng.integers(0, len(shape)))
    qmin, qmax = RANGES[int(rng.integers(0, len(RANGES)))]
    C = shape[ch_axis] if per_channel else 1
    # observer statistics (inf in the data makes min / max inf: finite qparams are drawn separately)
    mn, mx, _ = ops.channel_stats(x, ch_axis, per_channel)
    omn, omx = oracle.minmax(xf, ch_axis, per_channel)
    assert same_values(mn.cpu().numpy(), omn) and same_values(mx.cpu().numpy(), omx), (shape, ch_axis, per_channel)
    scale = np.abs(rng.standard_normal(C)).astype(np.float32) * 0.1 + 1e-3
    if seed % 7 == 0:
        scale[0] = 1e-6  # the observer's floor
    zp = np.zeros(C, np.float32) if qmin < 0 else rng.integers(0, qmax + 1, size=C).astype(np.float32)
    if seed % 5 == 0 and qmin >= 0:
        zp = zp + 0.5  # half-to-even rounding of the zero point
    ref_dq, ref_q = oracle.qdq(xf, scale, zp, qmin, qmax, ch_axis)
    y, q = ops.fake_quant(x, dev_tensor(scale), dev_tensor(zp), qmin, qmax, ch_axis, return_q=torch.int32)
    assert same_values(y.cpu().numpy(), ref_dq), (shape, ch_axis, per_channel, dtype, offset, qmin)
    assert np.array_equal(q.cpu().numpy(), ref_q)
    if dtype != torch.float32:
        y16 = ops.fake_quant(x, dev_tensor(scale), dev_tensor(zp), qmin, qmax, ch_axis, out_dtype=dtype)
        assert same_values(y16.float().cpu().numpy(), torch.from_numpy(ref_dq).to(dtype).float().numpy())
    # fused mask
    m = torch.from_numpy(rng.integers(0, 2, size=shape).astype(np.bool_)).cuda()
    ref_m, _ = oracle.qdq(xf, scale, zp, qmin, qmax, ch_axis, mask=m.cpu().numpy())
    ym = ops.fake_quant(x, dev_tensor(scale), dev_tensor(zp), qmin, qmax, ch_axis, mask=m)
    # inf * 0 is NaN in `w * mask` (sparse/modules/conv.py:40); the fused kernel selects -> 0: compare finite inputs
    finite = np.isfinite(xf)
    assert same_values(np.where(finite, ym.cpu().numpy(), 0), np.where(finite, ref_m, 0))
    # levels only / stored levels back to floats
    if qmax - qmin <= 255:
        qt = torch.int8 if qmin < 0 else torch.uint8
        q8 = ops.quantize_only(x, dev_tensor(scale), dev_tensor(zp), qmin, qmax, ch_axis, return_q=qt)
        assert np.array_equal(q8.cpu().numpy().astype(np.int32), ref_q)
        back = ops.dequantize_linear(q8, dev_tensor(scale), dev_tensor(zp), ch_axis=ch_axis)
        notnan = ~np.isnan(xf)  # an integer level cannot carry a NaN: those positions hold level 0
        assert same_values(np.where(notnan, back.cpu().numpy(), 0), np.where(notnan, ref_dq, 0))


@pytest.mark.parametrize("seed", range(30))
def test_fuzz_backward(oracle, seed):
    from sparsebit_amd import ops

    rng = np.random.default_rng(5000 + seed)
    shape = _shape(rng)
    dtype = DTYPES[seed % 3]
    offset = int(rng.choice([0, 0, 1, 4]))
    n = int(np.prod(shape))
    x = (torch.from_numpy(rng.standard_normal(n + offset).astype(np.float32)) * 2).to(dtype).cuda()[offset:].view(shape)
    gy = torch.from_numpy(rng.standard_normal(n + offset).astype(np.float32)).to(dtype).cuda()[offset:].view(shape)
    per_channel = bool(rng.integers(0, 2))
    ch_axis = int(rng.integers(0, len(shape)))
    qmin, qmax = RANGES[int(rng.integers(0, 5))]
    C = shape[ch_axis] if per_channel else 1
    scale = np.abs(rng.standard_normal(C)).astype(np.float32) * 0.2 + 0.05
    zp = np.zeros(C, np.float32) if qmin < 0 else rng.integers(0, qmax + 1, size=C).astype(np.float32)
    gx, gs, gz = ops.fake_quant_backward(x, gy, dev_tensor(scale), dev_tensor(zp), qmin, qmax, ch_axis)
    ogx, ogs, ogz = oracle.ste_backward(x.float().cpu().numpy(), gy.float().cpu().numpy(), scale, zp, qmin, qmax, ch_axis)
    assert same_values(gx.float().cpu().numpy(), torch.from_numpy(ogx).to(dtype).float().numpy()), (shape, ch_axis, dtype)
    tol = 2e-5 * max(1.0, float(np.abs(ogs).max()))
    assert np.allclose(gs.cpu().numpy(), ogs, rtol=1e-5, atol=tol)
    tolz = 2e-5 * max(1.0, float(np.abs(ogz).max()))
    assert np.allclose(gz.cpu().numpy(), ogz, rtol=1e-5, atol=tolz)


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_group_launches(seed):
    """random mixes of weight shapes / ranges / granularities through the model-wide launches == one by one"""
    from sparsebit_amd import ops

    rng = np.random.default_rng(9000 + seed)
    dtype = DTYPES[seed % 3]
    masked = bool(seed % 2)
    entries, masks, gys, lsq, ratios = [], [], [], [], []
    for _ in range(int(rng.integers(1, 12))):
        C = int(rng.choice([1, 2, 3, 8, 17, 64, 130]))
        inner = 8 * int(rng.choice([1, 2, 3, 9, 33, 64, 72, 300]))
        tail = (int(rng.choice([1, 3])),) if rng.integers(0, 3) == 0 and inner % 3 == 0 else ()
        shape = (C, inner // (tail[0] if tail else 1)) + tail
        x = (torch.from_numpy(rng.standard_normal(shape).astype(np.float32)) * 1.5).to(dtype).cuda()
        per_channel = bool(rng.integers(0, 4))
        qmin, qmax = RANGES[int(rng.integers(0, 5))]
        k = C if per_channel else 1
        scale = torch.from_numpy((np.abs(rng.standard_normal(k)) * 0.1 + 0.02).astype(np.float32)).cuda()
        is_lsq = bool(rng.integers(0, 2))
        if is_lsq:
            scale = scale * torch.from_numpy(rng.choice([-1.0, 1.0], size=k).astype(np.float32)).cuda()
        zp = torch.zeros(k).cuda() if qmin < 0 else torch.from_numpy(rng.integers(0, qmax + 1, size=k).astype(np.float32)).cuda()
        entries.append((x, scale, zp, qmin, qmax))
        masks.append(torch.from_numpy(rng.integers(0, 2, size=shape).astype(np.bool_)).cuda() if masked else None)
        gys.append(torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).to(dtype).cuda())
        lsq.append(is_lsq)
        ratios.append(float(rng.uniform(0.01, 1.0)) if is_lsq else 1.0)
    mk = masks if masked else None
    outs = ops.GroupFakeQuant(entries, masks=mk, lsq=lsq, fresh_outputs=bool(seed % 3))()
    gxs, gss = ops.GroupFakeQuantBackward(entries, masks=mk, lsq=lsq, want_gs=True, gs_ratios=ratios)(gys)
    for i, (x, scale, zp, qmin, qmax) in enumerate(entries):
        s_eff = scale.abs() if lsq[i] else scale
        z_eff = zp.clamp(qmin, qmax) if lsq[i] else zp
        assert torch.equal(outs[i], ops.fake_quant(x, s_eff, z_eff, qmin, qmax, 0, mask=masks[i])), i
        xm = x if not masked else x * masks[i]
        gx, gs, _ = ops.fake_quant_backward(xm, gys[i], s_eff, z_eff, qmin, qmax, 0, True, False)
        if masked:
            gx = gx * masks[i]
        assert torch.equal(gxs[i], gx), i
        if lsq[i]:
            gs = gs * ratios[i] * torch.sign(scale)
        assert torch.allclose(gss[i], gs, rtol=1e-5, atol=2e-5 * max(1.0, float(gs.abs().max()))), i


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_selection(oracle, seed):
    """percentile observer (row kernel and radix path, several cached batches) and the L1 mask threshold on
    random geometries with heavy duplicates, all-positive / all-negative channels and signed zeros"""
    from sparsebit_amd.config import quantizer_config, sparser_config
    from sparsebit_amd.observers import build_observer
    from sparsebit_amd.quantizers.quant_descriptor import QuantDescriptor
    from sparsebit_amd.sparsers import build_sparser

    rng = np.random.default_rng(20000 + seed)
    dtype = DTYPES[seed % 3]
    perch = bool(seed % 2)
    ch_axis, layout = [(0, None), (1, "NCHW"), (2, "NLC")][seed % 3 if perch else int(rng.integers(0, 3))]
    if layout is None:
        shape = (int(rng.choice([1, 3, 16, 40])), int(rng.choice([1, 7, 64, 1000, 5000])))
    elif layout == "NCHW":
        shape = (int(rng.choice([1, 2, 5])), int(rng.choice([1, 3, 8])), int(rng.choice([1, 7, 14])), int(rng.choice([1, 6, 9])))
    else:
        shape = (int(rng.choice([1, 2, 4])), int(rng.choice([1, 13, 50])), int(rng.choice([1, 8, 24, 65])))
    nb = 1 if layout is None else int(rng.integers(1, 4))
    xs = []
    for _ in range(nb):
        a = rng.standard_normal(shape).astype(np.float32)
        a[rng.random(shape) < 0.3] = float(rng.choice([0.0, -0.0, 0.25, -1.5]))  # duplicates, signed zeros
        if rng.random() < 0.3:
            a = np.abs(a)
        elif rng.random() < 0.2:
            a = -np.abs(a) - 0.5
        xs.append(torch.from_numpy(a).to(dtype))
    alpha = float(rng.choice([0.0, 1e-3, 0.01, 0.2, 0.5, 1.0]))
    name = "per-%s-affine" % ("channel" if perch else "tensor")
    cfg = quantizer_config(name, 8, observer="PERCENTILE", target="weight" if layout is None else "feature",
                           layout=layout or "NCHW", alpha=alpha)
    obs = build_observer(cfg, QuantDescriptor(cfg))
    for x in xs:
        obs.data_cache.update(x.cuda())
    mn, mx = obs.calc_minmax()
    if perch:
        data = np.concatenate([np.moveaxis(x.float().numpy(), ch_axis, 0).reshape(x.shape[ch_axis], -1) for x in xs], 1)
        rmn, rmx = oracle.percentile(data, alpha, 0, True)
    else:
        rmn, rmx = oracle.percentile(np.concatenate([x.float().numpy().reshape(-1) for x in xs]), alpha, per_channel=False)
    assert same_values(mn.reshape(-1).cpu().numpy(), rmn) and same_values(mx.reshape(-1).cpu().numpy(), rmx), (shape, alpha, perch)
    # mask threshold / mask of the first batch
    ratio = float(rng.choice([0.05, 0.5, 0.9, 0.999]))
    sp = build_sparser(sparser_config(ratio))
    w = xs[0].cuda()
    rm, rt = oracle.l1_mask(xs[0].float().numpy(), ratio)
    assert float(sp.calc_threshold(w)) == float(rt)
    assert np.array_equal(sp.calc_mask(w).cpu().numpy(), rm)


@pytest.mark.parametrize("seed", range(18))
def test_fuzz_gptq(oracle, seed):
    """random (batch, in, out, group, bits) incl. ragged sizes: every dispatch path of the mat-vec"""
    from sparsebit_amd import gptq

    rng = np.random.default_rng(30000 + seed)
    bit = [4, 3, 2][seed % 3]
    B = int(rng.choice([1, 2, 3, 5, 8, 17]))
    GS = int(rng.choice([-1, 128, 256] + ([64] if bit == 2 else [])))
    if GS == -1:
        M = int(rng.choice([8, 40, 136, 500, 1000, 4100]))
    else:
        M = GS * int(rng.integers(1, 9))
    N = int(rng.choice([1, 4, 33, 64, 96, 100, 256, 1027]))
    torch.manual_seed(seed)
    layer = torch.nn.Linear(M, N)
    x = torch.randn(B, M)
    w = layer.weight.data.numpy()
    scale, zero = oracle.gptq_find_params(w, bit, GS)
    wq = oracle.gptq_quantize(w, scale, zero, bit)
    qw, zeros_p = oracle.gptq_pack(wq, scale, zero, bit)
    ql = gptq.QuantLinear(M, N, bit=bit, groupsize=GS)
    ql.qweight = torch.from_numpy(qw)
    ql.scales = torch.from_numpy(scale).reshape(ql.scales.shape)
    ql.zeros = torch.from_numpy(zeros_p).reshape(ql.zeros.shape)
    ql.bias = layer.bias.detach().clone()
    ql = ql.cuda()
    y = ql(x.cuda())
    want = x.double() @ torch.from_numpy(wq).double().t() + layer.bias.detach().double()
    assert torch.allclose(y.double().cpu(), want, rtol=1e-5, atol=3e-5), (bit, B, M, N, GS, (y.double().cpu() - want).abs().max())
    assert torch.equal(y, ql(x.cuda()))
    ref = oracle.vecquantmatmul(x.numpy(), qw, layer.bias.detach().numpy(), scale, zeros_p, GS, bit)
    assert np.allclose(y.cpu().numpy(), ref, rtol=1e-5, atol=3e-5)


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_group_kth_value(seed):
    """round 6: the grouped whole-tensor selection (fp32: one launch, candidate store in LDS; 16-bit: the full-histogram /
    windowed engines per item) on random item counts, sizes (ragged tails, sub-slab items, items of many slabs), ranks
    (both ends, the bulk) and data (ties, two-valued, sorted runs -- the candidate store's overflow path --, outliers,
    signed zeros, NaN), plain and |x|: every item against a sort"""
    from sparsebit_amd import ops

    rng = np.random.default_rng(61000 + seed)
    dtype = DTYPES[seed % 3]
    n_items = int(rng.choice([1, 2, 5, 17, 70]))
    xs, ks = [], []
    for i in range(n_items):
        n = int(rng.choice([8, 9, 63, 1000, 16384, 16385, 40001, 131072 + 5, 300000, 1 << 20]))
        if n_items > 20 and n > 131077:
            n = 40001
        a = rng.standard_normal(n).astype(np.float32) * float(rng.choice([1e-3, 1.0, 40.0]))
        kind = int(rng.integers(0, 7))
        if kind == 0:
            a = np.sort(a)
        elif kind == 1:
            a[rng.random(n) < 0.5] = float(rng.choice([0.0, -0.0, 0.25]))
        elif kind == 2:
            a = np.where(rng.random(n) < 0.5, np.float32(1.0), np.float32(-1.0))
        elif kind == 3:
            a[rng.integers(0, n, size=max(1, n // 5000))] = np.float32(3e20) * np.float32(rng.choice([-1.0, 1.0]))
        elif kind == 4 and n > 100:
            a[rng.integers(0, n, size=3)] = np.float32("nan")
        t = torch.from_numpy(a).to(dtype)
        xs.append(t)
        ks.append(int(rng.choice([1, 2, n, max(n - 1, 1), max(n // 2, 1), max(n // 1000, 1), max((n * 9) // 10, 1)])))
    xd = [x.cuda() for x in xs]
    for use_abs in (False, True):
        got = ops.group_kth_value(xd, ks, use_abs).cpu().numpy()
        for i, x in enumerate(xs):
            a = x.float().numpy()
            want = np.sort(np.abs(a) if use_abs else a, kind="stable")[ks[i] - 1]  # (NaN last, as torch.sort)
            assert got[i] == want or (np.isnan(got[i]) and np.isnan(want)), (seed, i, x.numel(), ks[i], use_abs, got[i], want)


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_group_mse(oracle, seed):
    """round 6: the model-wide MSE search (a lane per (row, candidate); rows longer than 16 384 elements on the chunk form)
    on random tensor lists, row counts that are not multiples of four, row lengths 8 ... 20 000, symmetric / affine,
    4 / 8 bit: scale / zero point / index of every row against the oracle, ties of two candidates' losses within fp32
    rounding allowed (oracle.mse_index_disagreements)"""
    from sparsebit_amd import ops

    rng = np.random.default_rng(62000 + seed)
    dtype = DTYPES[seed % 3]
    symmetric = bool(seed % 2)
    qmin, qmax = [(-128, 127), (-8, 7)][seed % 2] if symmetric else [(0, 255), (0, 15)][(seed // 2) % 2]
    ws = []
    for _ in range(int(rng.integers(1, 9))):
        C = int(rng.choice([1, 2, 3, 5, 8, 33, 130]))
        inner = 8 * int(rng.choice([1, 2, 9, 64, 72, 128, 300, 576, 2049, 2500]))
        if C * inner > 400000:
            C = max(1, 400000 // inner)
        a = rng.standard_normal((C, inner)).astype(np.float32) * float(rng.choice([0.02, 1.0, 25.0]))
        if rng.random() < 0.3:
            a[int(rng.integers(0, C))] = 0.0
        if rng.random() < 0.3:
            a[int(rng.integers(0, C)), int(rng.integers(0, inner))] = 500.0
        ws.append(torch.from_numpy(a).to(dtype))
    wd = [w.cuda() for w in ws]
    grp = ops.GroupCalibration([(w, qmin, qmax, symmetric, True) for w in wd])
    s, z, idx = grp.mse_qparams()
    torch.cuda.synchronize()
    for i, w in enumerate(ws):
        rows = w.float().numpy()
        so, zo, bo, _ = oracle.mse(rows, qmin, qmax, symmetric, 0, True)
        got = idx[i].cpu().numpy()
        assert oracle.mse_index_disagreements(rows, got, bo, qmin, qmax, symmetric) == [], (seed, i, rows.shape)
        eq = got == bo
        assert np.array_equal(s[i].cpu().numpy()[eq], so[eq]) and np.array_equal(z[i].cpu().numpy()[eq], zo[eq]), (seed, i)
