"""Round-3 kernels against the oracle (all through the C ABI / ops.py):
  * the min-max-only statistics kernel (packed 16-bit integer reductions, v_minimum3 / v_maximum3 for fp32),
  * selections over more than 64 cached batches (the windowed engine's table holds 64),
  * the one-launch selection engine for 16-bit inputs,
  * model-wide calibration launches (statistics, thresholds, MSE) == the per-tensor calls, bit for bit,
  * the multi-matrix GPTQ mat-vec == per-matrix calls.
"""
import numpy as np
import pytest
import torch

from tests.helpers import same_values

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.float16, torch.bfloat16]


@pytest.fixture(scope="module")
def ops():
    from sparsebit_amd import ops as _ops

    return _ops


# --------------------------------------------------------------------------------------
# min-max observer kernel (observers/minmax.py:14-25)
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape,ch_axis", [((64, 4096), 0), ((37, 2048), 0), ((5, 8192 + 24), 0), ((1, 70001), 0),
                                           ((16, 3, 7, 7), 0), ((8, 40), 0), ((4, 16, 14, 14), 1), ((3, 100, 64), 2)])
def test_minmax_only_stats_vs_oracle(oracle, ops, shape, ch_axis, dtype):
    g = torch.Generator().manual_seed(7)
    x = (torch.randn(*shape, generator=g) * 3).to(dtype)
    xf = x.float().numpy()
    for perch in (True, False):
        mn, mx, ab = ops.channel_stats(x.cuda(), ch_axis, perch)  # min / max only: abssum not requested
        assert ab is None
        rmn, rmx = oracle.minmax(xf, ch_axis, perch)
        assert np.array_equal(mn.cpu().numpy(), rmn) and np.array_equal(mx.cpu().numpy(), rmx), (shape, perch)


@pytest.mark.parametrize("dtype", DTYPES)
def test_minmax_only_sign_and_special_rows(ops, dtype):
    """rows built to hit every branch of the (A, B, C) decode: all negative, all positive, only -0, only +0, mixed,
    +-inf, NaN of either sign, NaN with infinities, a single element that differs"""
    n = 4096
    base = torch.linspace(0.5, 3.0, n)
    rows = {
        "all_neg": -base,
        "all_pos": base,
        "neg_zero_only": torch.full((n,), -0.0),
        "pos_zero_only": torch.zeros(n),
        "mixed": torch.cat([base[: n // 2], -base[n // 2:]]),
        "neg_and_pos_zero": torch.cat([torch.full((n // 2,), -0.0), torch.zeros(n // 2)]),
        "pos_inf": torch.cat([base[:-1], torch.tensor([float("inf")])]),
        "neg_inf": torch.cat([-base[:-1], torch.tensor([float("-inf")])]),
        "both_inf": torch.cat([base[:-2], torch.tensor([float("inf"), float("-inf")])]),
        "pos_nan": torch.cat([base[:-1], torch.tensor([float("nan")])]),
        "neg_nan_all_neg": torch.cat([-base[:-1], -torch.tensor([float("nan")])]),
        "nan_with_inf": torch.cat([base[:-3], torch.tensor([float("inf"), float("nan"), float("-inf")])]),
        "one_negative": torch.cat([base[:-1], torch.tensor([-7.0])]),
        "one_positive": torch.cat([-base[:-1], torch.tensor([7.0])]),
        "tiny": torch.cat([torch.full((n - 2,), 1e-6), torch.tensor([-1e-7, 2e-7])]),
    }
    x = torch.stack(list(rows.values())).to(dtype)
    # a negative NaN must survive the cast with its sign: build it from bits
    if dtype != torch.float32:
        bits = x.view(torch.int16)
        k = list(rows).index("neg_nan_all_neg")
        bits[k, -1] = torch.tensor(-1, dtype=torch.int16)  # 0xffff: a negative NaN in both 16-bit formats
    xf = x.float()
    want_mn = torch.where(torch.isnan(xf).any(1), torch.tensor(float("nan")), xf.min(1).values).numpy()
    want_mx = torch.where(torch.isnan(xf).any(1), torch.tensor(float("nan")), xf.max(1).values).numpy()
    mn, mx, _ = ops.channel_stats(x.cuda(), 0, True)
    assert same_values(mn.cpu().numpy(), want_mn), list(zip(rows, mn.cpu().tolist(), want_mn.tolist()))
    assert same_values(mx.cpu().numpy(), want_mx), list(zip(rows, mx.cpu().tolist(), want_mx.tolist()))
    # per tensor over the same data: several partial records folded by the finish kernel
    mn, mx, _ = ops.channel_stats(x[:8].contiguous().cuda(), 0, False)
    assert mn.item() == xf[:8].min().item() and mx.item() == xf[:8].max().item()


def test_minmax_only_equals_general_kernel(ops):
    """knob 2 == 11 routes the same call through the general (min, max, sum|x|) kernel"""
    from sparsebit_amd import lib as L

    g = torch.Generator().manual_seed(3)
    for dtype in DTYPES:
        x = (torch.randn(512, 4096, generator=g) * torch.logspace(-3, 2, 512).unsqueeze(1)).to(dtype).cuda()
        a = ops.channel_stats(x, 0, True)
        try:
            L.set_tuning(2, 11)
            b = ops.channel_stats(x, 0, True)
        finally:
            L.set_tuning(2, 0)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


# --------------------------------------------------------------------------------------
# selections over many cached batches (the reference accepts any number: observers/base.py:12-36)
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_percentile_over_more_than_64_batches(oracle, ops, dtype):
    g = torch.Generator().manual_seed(65)
    batches = [(torch.randn(4, 33, 64, generator=g) * (1 + i % 5)).to(dtype) for i in range(70)]
    mn, mx = ops.percentile_select([b.cuda() for b in batches], 1e-2, 0, False)
    flat = np.concatenate([b.float().numpy().reshape(-1) for b in batches])
    rmn, rmx = oracle.percentile(flat, 1e-2, 0, False)
    assert np.array_equal(mn.cpu().numpy().reshape(-1), rmn) and np.array_equal(mx.cpu().numpy().reshape(-1), rmx)
