"""CPU: pin the oracle (oracle/sbq_oracle.c) against the real reference's outputs
(tests/golden/ref_golden.npz) and the hand-checked KATs of SURVEY.md 8c."""
import numpy as np
import pytest

from conftest import golden_cases, rel_err


def _meta(golden, name):
    qmin, qmax, ch_axis, perch, sym = golden[name + "/meta"].tolist()
    return int(qmin), int(qmax), int(ch_axis), bool(perch), bool(sym)


def _all_x(golden, name):
    xs = [golden[name + "/x"]]
    i = 1
    while name + "/x%d" % i in golden:
        xs.append(golden[name + "/x%d" % i])
        i += 1
    return xs


QUANT_CASES = golden_cases("")


@pytest.mark.parametrize("name", QUANT_CASES)
def test_qdq_matches_reference(golden, oracle, name):
    """oracle QDQ(x, ref scale, ref zp) == reference dq, bit for bit (zeros by value)."""
    qmin, qmax, ch_axis, perch, sym = _meta(golden, name)
    x = golden[name + "/x"]
    dq, q = oracle.qdq(x, golden[name + "/scale"], golden[name + "/zero_point"], qmin, qmax, ch_axis)
    ref = golden[name + "/dq"]
    assert np.array_equal(dq, ref), "max rel err %g" % rel_err(dq, ref)
    assert q.min() >= qmin and q.max() <= qmax


@pytest.mark.parametrize("name", [c for c in QUANT_CASES if c.split("/")[0] in ("uni", "trt", "act")])
def test_minmax_observer_matches_reference(golden, oracle, name):
    qmin, qmax, ch_axis, perch, sym = _meta(golden, name)
    xs = _all_x(golden, name)
    if perch:
        assert len(xs) == 1
        mn, mx = oracle.minmax(xs[0], ch_axis, True)
    else:
        mns, mxs = zip(*[oracle.minmax(x, ch_axis, False) for x in xs])
        mn, mx = np.min(mns, axis=0), np.max(mxs, axis=0)  # min/max over shards is exact
    assert np.array_equal(mn, golden[name + "/min_val"])
    assert np.array_equal(mx, golden[name + "/max_val"])
    s, z = oracle.qparams_from_minmax(mn, mx, qmin, qmax, sym)
    assert np.array_equal(s, golden[name + "/scale"])
    assert np.array_equal(z, golden[name + "/zero_point"])


@pytest.mark.parametrize("name", golden_cases("pct/"))
def test_percentile_observer_matches_reference(golden, oracle, name):
    qmin, qmax, ch_axis, perch, sym = _meta(golden, name)
    alpha = float(name.split("/")[2])
    xs = _all_x(golden, name)
    if perch:
        data = xs[0]
        mn, mx = oracle.percentile(data, alpha, ch_axis, True)
    else:
        data = np.concatenate([x.reshape(-1) for x in xs])  # DataCache LAYERWISE, base.py:32-33
        mn, mx = oracle.percentile(data, alpha, per_channel=False)
    assert np.array_equal(mn, golden[name + "/min_val"])
    assert np.array_equal(mx, golden[name + "/max_val"])
    s, z = oracle.qparams_from_minmax(mn, mx, qmin, qmax, sym)
    assert np.array_equal(s, golden[name + "/scale"])
    assert np.array_equal(z, golden[name + "/zero_point"])


@pytest.mark.parametrize("name", golden_cases("mse/"))
def test_mse_observer_matches_reference(golden, oracle, name):
    """per-tensor MSE: the reference's chosen candidate (its fp32 loss order) must be the
    oracle's (fp64 accumulation) -- scale and zero_point bit-equal."""
    qmin, qmax, ch_axis, perch, sym = _meta(golden, name)
    assert not perch
    xs = _all_x(golden, name)
    # mse.py:29 uses the CHANNELWISE cat even per tensor; for loss/minmax only the multiset matters
    data = np.concatenate([x.reshape(-1) for x in xs])
    s, z, best, sse = oracle.mse(data, qmin, qmax, sym, per_channel=False)
    assert np.array_equal(s, golden[name + "/scale"]), (best, s, golden[name + "/scale"])
    assert np.array_equal(z, golden[name + "/zero_point"])


@pytest.mark.parametrize("name", golden_cases("lsq/"))
def test_lsq_init_matches_reference(golden, oracle, name):
    qmin, qmax, ch_axis, perch, sym = _meta(golden, name)
    x = golden[name + "/x"]
    s = oracle.lsq_init_scale(x, qmax, ch_axis, perch)
    assert rel_err(s, golden[name + "/scale"]) <= 1e-6  # fp32 vs fp64 mean: tolerance of the contract
    assert np.all(golden[name + "/zero_point"] == 0)


def test_hand_kats(golden, oracle):
    x = golden["kat/x"]
    for key, s, zp, lo, hi in (("kat/int8_s1_zp0", 1.0, 0.0, -128, 127), ("kat/uint8_s1_zp3.5", 1.0, 3.5, 0, 255),
                                ("kat/uint8_s0.3_zp2.5", 0.3, 2.5, 0, 255)):
        dq, q = oracle.qdq(x, [s], [zp], lo, hi)
        assert np.array_equal(dq, golden[key]), key
    # SURVEY.md 8c, first 11 values
    dq, q = oracle.qdq(x[:11], [1.0], [0.0], -128, 127)
    assert q.tolist() == [0, 2, 2, 0, -2, -2, 126, 127, 127, -128, -128]
    dq, q = oracle.qdq(np.array([127.5, 200, -128.5], np.float32), [1.0], [3.5], 0, 255)
    assert dq.tolist() == [128.0, 200.0, -4.0]
    dq, _ = oracle.qdq(golden["kat/tie_x"], golden["kat/tie_s"], [0.0], -128, 127)
    assert np.array_equal(dq, golden["kat/tie_dq"])
    # calc_qparams_with_minmax KATs
    s, z = oracle.qparams_from_minmax([-1], [3], -128, 127, True)
    assert s[0] == np.float32(6.0) / np.float32(255.0) and z[0] == 0
    s, z = oracle.qparams_from_minmax([-1], [3], 0, 255, False)
    assert s[0] == np.float32(4.0) / np.float32(255.0) and z[0] == 64
    s, z = oracle.qparams_from_minmax([1], [3], 0, 255, False)
    assert s[0] == np.float32(3.0) / np.float32(255.0) and z[0] == 0
    s, z = oracle.qparams_from_minmax([0], [0], 0, 255, False)
    assert s[0] == np.float32(1e-6)
    mn, mx = oracle.percentile(np.arange(-1000, 1000, dtype=np.float32), 1e-3, per_channel=False)
    assert (mn[0], mx[0]) == (-1000.0, 998.0)


@pytest.mark.parametrize("ratio", [0.0, 0.3, 0.5, 0.9, 1.0])
@pytest.mark.parametrize("wname", ["lin", "conv"])
def test_l1_mask_matches_reference(golden, oracle, ratio, wname):
    x = golden["uni/per-channel-symmetric/8/%s/x" % wname]
    m, t = oracle.l1_mask(x, ratio)
    assert np.array_equal(m.astype(np.uint8), golden["mask/%g/%s" % (ratio, wname)])


def test_l1_mask_kat_and_masked_lsq(golden, oracle):
    m, t = oracle.l1_mask(golden["mask/kat_x"], 0.5)
    assert m.astype(int).tolist() == [[0, 0, 0, 1], [0, 1, 1, 0]] and t == 2.0
    assert np.array_equal(m.astype(np.uint8), golden["mask/kat"])
    x = golden["maskq/x"]
    dq, q = oracle.qdq(x, np.abs(golden["maskq/scale"]), np.zeros_like(golden["maskq/scale"]), -8, 7, 0,
                       mask=golden["maskq/mask"])
    assert np.array_equal(dq, golden["maskq/dq"])


@pytest.mark.parametrize("name", ["bwd/pc4", "bwd/pt8a", "bwd/pc8a_nchw"])
def test_ste_backward_matches_reference(golden, oracle, name):
    qmin, qmax, ch_axis = [int(v) for v in golden[name + "/meta"]]
    gx, gs, gz = oracle.ste_backward(golden[name + "/x"], golden[name + "/gy"], golden[name + "/scale"],
                                     golden[name + "/zero_point"], qmin, qmax, ch_axis)
    assert np.array_equal(gx, golden[name + "/gx"])
    # reduction order differs (reference elementwise fp32 products summed in fp64 here too)
    assert np.allclose(gs, golden[name + "/gs"], rtol=1e-5, atol=1e-5)
    assert np.allclose(gz, golden[name + "/gzp"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", ["g128", "g-1", "rag"])
def test_gptq_matches_reference(golden, oracle, name):
    B, M, N, GS = [int(v) for v in golden["gptq/%s/meta" % name]]
    w = golden["gptq/%s/w" % name]
    scale, zero = oracle.gptq_find_params(w, 4, GS)
    assert np.array_equal(scale, golden["gptq/%s/scale" % name])
    assert np.array_equal(zero, golden["gptq/%s/zero" % name])
    wq = oracle.gptq_quantize(w, scale, zero)
    assert np.array_equal(wq, golden["gptq/%s/wq" % name])
    qw, zeros_p = oracle.gptq_pack4(wq, scale, zero)
    assert np.array_equal(qw, golden["gptq/%s/qweight" % name])
    assert np.array_equal(zeros_p, golden["gptq/%s/zeros" % name])
    y = oracle.vecquant4matmul(golden["gptq/%s/x" % name], qw, golden["gptq/%s/bias" % name],
                               golden["gptq/%s/scales" % name], zeros_p, GS)
    # the reference's own tolerance (test_cuda_kernel.py:45)
    assert np.allclose(y, golden["gptq/%s/y" % name], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", ["b3/g128", "b3/g-1", "b3/rag", "b3/strip", "b2/g64", "b2/g-1", "b2/rag", "b2/strip"])
def test_gptq_low_bit_matches_reference(golden, oracle, name):
    """3- and 2-bit find_params / quantize / pack / mat-vec against the reference's own outputs."""
    B, M, N, GS, bit = [int(v) for v in golden["gptq/%s/meta" % name]]
    w = golden["gptq/%s/w" % name]
    scale, zero = oracle.gptq_find_params(w, bit, GS)
    assert np.array_equal(scale, golden["gptq/%s/scale" % name])
    assert np.array_equal(zero, golden["gptq/%s/zero" % name])
    wq = oracle.gptq_quantize(w, scale, zero, bit)
    assert np.array_equal(wq, golden["gptq/%s/wq" % name])
    qw, zeros_p = oracle.gptq_pack(wq, scale, zero, bit)
    assert qw.shape == golden["gptq/%s/qweight" % name].shape
    assert np.array_equal(qw, golden["gptq/%s/qweight" % name])
    assert np.array_equal(zeros_p, golden["gptq/%s/zeros" % name])
    y = oracle.vecquantmatmul(golden["gptq/%s/x" % name], qw, golden["gptq/%s/bias" % name],
                              golden["gptq/%s/scales" % name], zeros_p, GS, bit)
    assert np.allclose(y, golden["gptq/%s/y" % name], rtol=1e-5, atol=1e-5)


def test_gptq_pack_generic_equals_pack4(golden, oracle):
    for name in ("g128", "g-1", "rag"):
        wq, scale, zero = (golden["gptq/%s/%s" % (name, k)] for k in ("wq", "scale", "zero"))
        a, za = oracle.gptq_pack4(wq, scale, zero)
        b, zb = oracle.gptq_pack(wq, scale, zero, 4)
        assert np.array_equal(a, b) and np.array_equal(za, zb)


# ---- round 2: per-channel MSE pinned to the reference itself (row by row) ---------------------------
def _r02():
    import os

    from conftest import ROOT

    return np.load(os.path.join(ROOT, "tests", "golden", "ref_golden_r02.npz"), allow_pickle=False)


def _r02_cases(prefix):
    return [c for c in _r02()["cases"].tolist() if c.startswith(prefix)]


@pytest.mark.parametrize("name", _r02_cases("rowmse/"))
def test_oracle_perchannel_mse_equals_reference_row_by_row(oracle, name):
    """observers/mse.py:28-63 run by the REFERENCE on every row as its own per-tensor problem
    (tests/golden/gen_golden_r02.py) == the oracle's per-channel restatement: scale bit for bit."""
    z = _r02()
    wname = name.split("/")[-1]
    x = z["rowmse/%s/x" % wname]
    qmin, qmax, sym = [int(v) for v in z[name + "/meta"]]
    s, zp, best, _ = oracle.mse(x, qmin, qmax, bool(sym), 0, True)
    assert np.array_equal(s, z[name + "/scale"]), np.flatnonzero(s != z[name + "/scale"])
    assert np.array_equal(zp, z[name + "/zero_point"])


def test_oracle_gptq_find_params_sym_and_mse_vs_reference_golden(oracle):
    """round-3 goldens (tests/golden/gen_golden_r03.py): Quantizer.find_params with sym / mse from the reference itself.
    The grid search may pick another candidate only where the REFERENCE's own errors of the two tie to rounding
    (numpy's pow and summation order differ from torch's in the last bits)."""
    import os

    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_golden_r03.npz"), allow_pickle=False)
    for name in z["cases"].tolist():
        _, wname, bit, sym, gs = name.split("/")
        bit, gs = int(bit[1:]), int(gs[1:])
        w = z["gptqmse/%s/w" % wname]
        scale, zero, errs = oracle.gptq_find_params(w, bit, gs, sym=sym == "sym", mse=True)
        s_ref, z_ref, e_ref = z[name + "/scale"], z[name + "/zero"], z[name + "/errs"]
        assert np.allclose(errs, e_ref, rtol=2e-5, atol=0), name
        bad = np.nonzero((scale.reshape(-1) != s_ref) | (zero.reshape(-1) != z_ref))[0]
        for r in bad:
            mine = int(np.argmin(errs[r]))
            assert abs(e_ref[r, mine] - e_ref[r].min()) <= 2e-5 * abs(e_ref[r].min()), (name, r)
        assert len(bad) <= max(1, len(s_ref) // 20), (name, len(bad))
        # without the search: exact
        s0, z0 = oracle.gptq_find_params(w, bit, gs, sym=sym == "sym", mse=False)
        if sym == "sym":
            assert np.all(z0 == (2 ** bit) / 2)


def test_mse_index_disagreements_allows_fp32_ties_only(oracle):
    """oracle.mse_index_disagreements (the gate of kernels whose summation tree differs from the per-tensor kernel's):
    equal indices pass, an index whose loss is NOT a tie of the reported one is a disagreement, a true tie (two
    candidates with identical sums: an all-zero row) passes"""
    rng = np.random.default_rng(3)
    rows = (rng.standard_normal((6, 256)) * 0.3).astype(np.float32)
    rows[4] = 0.0  # every candidate's loss is exactly 0: any index ties with any other
    _, _, best, sse = oracle.mse(rows, -128, 127, True, 0, True)
    assert oracle.mse_index_disagreements(rows, best, best, -128, 127, True) == []
    other = best.copy()
    far = int(np.argmax(sse[0]))  # the WORST candidate of row 0: not a tie
    assert sse[0, far] > sse[0, best[0]] * 1.01
    other[0] = far
    other[4] = (best[4] + 7) % 80 if best[4] >= 0 else 3
    got = oracle.mse_index_disagreements(rows, other, best, -128, 127, True)
    assert 0 in got and (4 not in got or best[4] < 0)
