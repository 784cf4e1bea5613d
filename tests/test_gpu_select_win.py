"""The sample-guided windowed selection (csrc/sbq_select_win.hip) of a whole tensor: k-th values and percentile
min / max, bit-exact against numpy's exact order statistics (np.partition on the same fp32 data, i.e. what
torch.kthvalue / torch.sort return -- percentile.py:16-46, l1norm.py:18-26), against the oracle, and against the
fixed-digit radix engine (knob 2 = 7).  The data include everything that defeats a sample: sorted and periodic
inputs whose period equals the sampling stride, two-valued and constant tensors, heavy ties at the target rank,
ranks at both ends, NaN / inf, tensors smaller than the sample and several cached batches.
"""
import numpy as np
import pytest
import torch

from helpers import same_values
from sparsebit_amd import lib as L
from sparsebit_amd import ops

pytestmark = pytest.mark.gpu

DTYPES = [torch.bfloat16, torch.float16, torch.float32]


def _kth_ref(xf, k, use_abs):
    a = np.abs(xf) if use_abs else xf
    a = a.reshape(-1)
    # NaN sorts last (torch.sort / kthvalue); np.partition does the same
    return np.partition(a, k - 1)[k - 1]


def _datasets(n, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    out = {}
    out["gauss"] = torch.randn(n, generator=g)
    out["sorted"] = torch.sort(torch.randn(n, generator=g))[0]
    out["reversed"] = torch.sort(torch.randn(n, generator=g), descending=True)[0]
    stride = max(n // 16384, 1)
    per = torch.randn(n, generator=g)
    per[::stride] = 1000.0  # every sampled element is an outlier: the sample sees a constant
    out["period_eq_stride"] = per
    out["two_values"] = (torch.rand(n, generator=g) < 0.5).float() * 2 - 1
    out["constant"] = torch.full((n,), 0.37)
    ties = torch.randn(n, generator=g)
    ties[: n // 2] = 0.25  # half of the tensor on one key
    out["half_tied"] = ties[torch.randperm(n, generator=g)]
    heavy = torch.randn(n, generator=g) * torch.exp(3 * torch.randn(n, generator=g))
    out["heavy_tail"] = heavy
    out["tiny_values"] = torch.randn(n, generator=g) * 1e-30
    return {k: v.to(dtype) for k, v in out.items()}


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n", [1, 7, 1000, 16384, 16385, 300001, 4096 * 1024 + 3])
def test_kth_value_all_datasets(dtype, n):
    for name, x in _datasets(n, dtype, n % 1000 + 1).items():
        xf = x.float().numpy()
        xd = x.cuda()
        ks = sorted({1, n, max(1, n // 2), max(1, n // 1000), max(1, n - n // 1000), min(n, int(n * 0.77) + 1)})
        for use_abs in (False, True):
            for k in ks:
                got = float(ops.kth_value(xd, k, use_abs))
                want = float(_kth_ref(xf, k, use_abs))
                assert got == want or (np.isnan(got) and np.isnan(want)), (name, n, k, use_abs, got, want)


@pytest.mark.parametrize("dtype", DTYPES)
def test_kth_value_equals_fixed_digit_engine(dtype):
    n = 2_000_003
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(n, generator=g) * torch.exp(torch.randn(n, generator=g))).to(dtype)
    x[100:110] = float("nan")
    x[200] = float("inf")
    x[201] = float("-inf")
    x[300:400] = -0.0
    xd = x.cuda()
    try:
        for use_abs in (False, True):
            for k in (1, 2, 1000, n // 2, n - 1000, n - 11, n - 10, n - 9, n):
                L.set_tuning(2, 7)
                old = ops.kth_value(xd, k, use_abs).cpu().numpy()
                L.set_tuning(2, 0)
                new = ops.kth_value(xd, k, use_abs).cpu().numpy()
                assert same_values(old, new), (k, use_abs, old, new)
    finally:
        L.set_tuning(2, 0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("alpha", [0.0, 1e-5, 1e-3, 0.01, 0.3, 0.5, 1.0])
def test_percentile_per_tensor_batches_vs_oracle(oracle, dtype, alpha):
    """percentile.py:16-46 over three cached batches (per tensor: one row of all their elements)."""
    g = torch.Generator().manual_seed(int(alpha * 1e6) + 3)
    for name, maker in (("gauss", lambda: torch.randn(64, 197, 96, generator=g)),
                        ("relu", lambda: torch.relu(torch.randn(64, 197, 96, generator=g))),   # no negatives
                        ("neg", lambda: -torch.rand(64, 197, 96, generator=g) - 0.1),            # no non-negatives
                        ("sorted", lambda: torch.sort(torch.randn(64 * 197 * 96, generator=g))[0].reshape(64, 197, 96))):
        xs = [maker().to(dtype) for _ in range(3)]
        mn, mx = ops.percentile_select([x.cuda() for x in xs], alpha, 0, False)
        data = np.concatenate([x.float().numpy().reshape(-1) for x in xs])
        rmn, rmx = oracle.percentile(data, alpha, per_channel=False)
        assert same_values(mn.cpu().numpy(), rmn) and same_values(mx.cpu().numpy(), rmx), (name, alpha, mn, rmn, mx, rmx)


def test_mask_threshold_config5_size(oracle):
    """l1norm.py:18-26 at the headline size: the 50 % threshold of a 4096x4096 bf16 weight."""
    g = torch.Generator().manual_seed(55)
    w = (torch.randn(4096, 4096, generator=g) * torch.logspace(-2, 1, 4096).unsqueeze(1)).bfloat16()
    n = w.numel()
    for ratio in (0.5, 0.9, 0.01):
        idx = min(int(n * ratio), n - 1)
        got = float(ops.kth_value(w.cuda(), idx + 1, True))
        want = float(_kth_ref(w.float().numpy(), idx + 1, True))
        assert got == want, (ratio, got, want)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,inner", [(4096, 4096), (513, 2048), (7, 96), (300, 4088)])
@pytest.mark.parametrize("alpha", [0.0, 1e-4, 1e-3, 1.9e-3])
def test_percentile_rows_small_ranks(oracle, dtype, C, inner, alpha):
    """the one-wave-per-row extraction path of sbq_percentile_rows (alpha * inner <= 8) against the oracle and
    against the general rows kernel (knob 2 = 5), with ties, NaNs and one-sided rows at the ends"""
    g = torch.Generator().manual_seed(C + inner)
    x = (torch.randn(C, inner, generator=g) * torch.logspace(-2, 1, C).unsqueeze(1)).to(dtype)
    x[0] = x[0].abs()                    # no negatives: min stays 0
    x[1] = -x[1].abs() - 0.01            # no non-negatives: max stays 0
    x[2, :5] = x[2].min()                # five copies of the minimum
    x[3, -6:] = x[3].max()               # six copies of the maximum
    x[4, 1] = float("nan")
    x[5, :] = 0.5                        # constant row
    x[6, :3] = float("-inf")
    xd = x.cuda()
    mn, mx = ops.percentile_rows(xd, alpha)
    rmn, rmx = oracle.percentile(x.float().numpy(), alpha, 0, True)
    assert same_values(mn.cpu().numpy(), rmn) and same_values(mx.cpu().numpy(), rmx)
    try:
        L.set_tuning(2, 5)
        gmn, gmx = ops.percentile_rows(xd, alpha)
    finally:
        L.set_tuning(2, 0)
    assert same_values(mn.cpu().numpy(), gmn.cpu().numpy()) and same_values(mx.cpu().numpy(), gmx.cpu().numpy())


@pytest.mark.parametrize("dtype", DTYPES)
def test_one_launch_over_mixed_shards(oracle, dtype):
    """the sweep is ONE launch over all cached batches: whole 16 Ki-element slabs of aligned shards take the lean
    path, ragged last slabs and every slab of a shard that is not 16-byte aligned the per-element path -- mix them:
    a large aligned shard, an unaligned view, a shard shorter than a slab, one element, a ragged multi-slab shard"""
    g = torch.Generator().manual_seed(77)
    big = torch.randn(3 * 16384 * 5 + 8, generator=g).to(dtype).cuda()
    base = torch.randn(70001, generator=g).to(dtype).cuda()
    shards = [big[: 3 * 16384 * 5], base[1:50000], torch.randn(777, generator=g).to(dtype).cuda(),
              torch.randn(1, generator=g).to(dtype).cuda(), torch.randn(2 * 16384 + 4099, generator=g).to(dtype).cuda()]
    assert shards[1].data_ptr() % 16 != 0
    data = np.concatenate([s.float().cpu().numpy().reshape(-1) for s in shards])
    # through the C ABI directly: shards of different sizes share the row length 1 (ops.percentile_select hands
    # such batches to the stepwise protocol instead)
    import ctypes
    lib = L.load()
    dev = shards[0].device
    outers = (ctypes.c_int64 * len(shards))(*[s.numel() for s in shards])
    ptrs = (ctypes.c_void_p * len(shards))(*[s.data_ptr() for s in shards])
    ws = torch.zeros(lib.sbq_radix_select_workspace_bytes(1, 2), dtype=torch.uint8, device=dev)  # zero before first use (sbq.h)
    for alpha in (0.0, 1e-3, 0.2):
        mn = torch.empty(1, dtype=torch.float32, device=dev)
        mx = torch.empty(1, dtype=torch.float32, device=dev)
        rc = lib.sbq_percentile_select(ptrs, outers, len(shards), L.dtype_id(shards[0]), 1, 1, float(alpha), L.ptr(mn),
                                       L.ptr(mx), L.ptr(ws), ws.numel(), L.stream_ptr(dev))
        L.check(rc)
        rmn, rmx = oracle.percentile(data, alpha, per_channel=False)
        assert same_values(mn.cpu().numpy(), rmn) and same_values(mx.cpu().numpy(), rmx), (alpha, mn, rmn, mx, rmx)
    # the mask threshold on single tensors of the same kinds: the unaligned view, the ragged multi-slab shard
    for t in (shards[1], shards[4]):
        tf = t.float().cpu().numpy()
        n = tf.size
        for k in (1, n // 2, n):
            assert float(ops.kth_value(t, k, True)) == float(_kth_ref(tf, k, True)), (n, k)
