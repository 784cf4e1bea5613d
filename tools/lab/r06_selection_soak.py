"""lab (round 6): a soak of the whole-tensor selections against torch.sort -- random sizes (8 ... 30 M), dtypes, data
kinds (ties, two values, sorted runs, outliers, zeros, NaN), ranks and percentiles, one tensor / several shards / grouped,
with the resignation knobs (2 = 31 / 32 / 33) mixed in.  Other seeds and far larger sizes than tests/test_gpu_fuzz.py.
  SOAK_SECONDS=240 python tools/lab/r06_selection_soak.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from sparsebit_amd import lib as L  # noqa: E402
from sparsebit_amd import ops  # noqa: E402

budget = float(os.environ.get("SOAK_SECONDS", "120"))
seed0 = int(os.environ.get("SOAK_SEED", "777"))
dev = torch.device("cuda:0")
DT = [torch.float32, torch.bfloat16, torch.float16]


def make(rng, n, dtype):
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(1 << 30)))
    a = torch.randn(n, generator=g, device=dev) * float(rng.choice([1e-3, 1.0, 40.0]))
    kind = int(rng.integers(0, 8))
    if kind == 0:
        a = torch.sort(a).values
    elif kind == 1:
        a[torch.rand(n, generator=g, device=dev) < 0.5] = float(rng.choice([0.0, -0.0, 0.25]))
    elif kind == 2:
        a = torch.where(torch.rand(n, generator=g, device=dev) < 0.5, 1.0, -1.0)
    elif kind == 3:
        a[torch.randint(0, n, (max(1, n // 5000),), generator=g, device=dev)] = 3e20 * float(rng.choice([-1.0, 1.0]))
    elif kind == 4 and n > 100:
        a[torch.randint(0, n, (3,), generator=g, device=dev)] = float("nan")
    elif kind == 5:
        a = torch.relu(a)
    elif kind == 6:
        a = torch.sort(a, descending=True).values
    return a.to(dtype).contiguous(), kind


def kth_ref(x, k, use_abs):
    v = x.float()
    v = v.abs() if use_abs else v
    return torch.sort(v).values[k - 1]  # (NaN last)


def pct_ref(xs, alpha):
    v = torch.cat([x.float().reshape(-1) for x in xs])
    n = v.numel()
    neg = int((v < 0).sum())
    nan = int(torch.isnan(v).sum())
    pos = n - neg - nan  # percentile.py:36-43 counts x >= 0 via (x >= 0).sum(); NaN compares false
    s = torch.sort(v).values
    kmin = max(round(neg * alpha), 1)
    kmax = n - max(round(pos * alpha), 0)
    kmax = min(max(kmax, 1), n)
    mn = s[kmin - 1] if neg > 0 else torch.zeros((), device=dev)
    mx = s[kmax - 1] if pos > 0 else torch.zeros((), device=dev)
    return mn, mx


def same(a, b):
    a, b = float(a), float(b)
    return a == b or (a != a and b != b)


t_end = time.time() + budget
it = fails = 0
counts = {"kth": 0, "pct": 0, "group": 0}
while time.time() < t_end:
    rng = np.random.default_rng(seed0 + it)
    it += 1
    L.set_tuning(2, int(rng.choice([0, 0, 0, 31, 32, 33])))
    dtype = DT[int(rng.integers(0, 3))]
    what = int(rng.integers(0, 3))
    if what == 0:
        n = int(rng.choice([8, 100, 16384, 16389, 65536, 1 << 20, 3_000_001, 16_777_216, 16_777_224, 25_600_000, 30_000_003]))
        x, kind = make(rng, n, dtype)
        k = int(rng.choice([1, 2, n, max(n - 1, 1), max(n // 2, 1), max(n // 1000, 1), max((n * 9) // 10, 1)]))
        ua = bool(rng.integers(0, 2))
        got, want = ops.kth_value(x, k, ua), kth_ref(x, k, ua)
        ok = same(got, want)
        desc = ("kth", str(dtype), n, kind, k, ua, float(got), float(want))
        counts["kth"] += 1
    elif what == 1:
        ns = int(rng.choice([1, 2, 4, 7]))
        inner = int(rng.choice([8, 1000, 16384, 75648, 1 << 20, 4_841_472, 6_000_008]))
        if ns * inner > 30_000_000:
            inner = 1 << 20
        xs = [make(rng, inner, dtype)[0] for _ in range(ns)]
        alpha = float(rng.choice([1e-5, 1e-3, 1e-2, 0.3]))
        mn, mx = ops.percentile_select(xs, alpha, 0, False)
        rmn, rmx = pct_ref(xs, alpha)
        ok = same(mn.reshape(-1)[0], rmn) and same(mx.reshape(-1)[0], rmx)
        desc = ("pct", str(dtype), ns, inner, alpha, float(mn.reshape(-1)[0]), float(rmn), float(mx.reshape(-1)[0]), float(rmx))
        counts["pct"] += 1
    else:
        n_items = int(rng.choice([1, 3, 20, 53]))
        xs, ks = [], []
        for _ in range(n_items):
            n = int(rng.choice([8, 4096, 16385, 147456, 589824, 2_359_296] if n_items > 3 else [16384, 2_359_296, 8_000_000]))
            xs.append(make(rng, n, dtype)[0])
            ks.append(int(rng.choice([1, n, max(n // 2, 1), max((n * 9) // 10, 1)])))
        ua = bool(rng.integers(0, 2))
        got = ops.group_kth_value(xs, ks, ua)
        ok = all(same(got[i], kth_ref(x, ks[i], ua)) for i, x in enumerate(xs))
        desc = ("group", str(dtype), n_items, [x.numel() for x in xs][:6], ks[:6], ua)
        counts["group"] += 1
    if os.environ.get("SOAK_VERBOSE") and it <= int(os.environ["SOAK_VERBOSE"]):
        print(it, ok, desc, flush=True)
    if not ok:
        fails += 1
        print("MISMATCH", it, "knob", L.get_tuning(2) if hasattr(L, "get_tuning") else "?", desc, flush=True)
L.set_tuning(2, 0)
print("selection soak: %d cases in %.0f s (%r), seed base %d, mismatches %d" % (it, budget, counts, seed0, fails), flush=True)
sys.exit(1 if fails else 0)
