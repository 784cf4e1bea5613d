"""Round 6: the k-th value of ONE fp32 tensor (the L1 mask threshold of an fp32 weight) -- the grouped selection's one-launch
form with a single item (candidate store in LDS) against round 3's three launches (knob 2 = 34)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from sparsebit_amd import lib as L  # noqa: E402
from sparsebit_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for n in (4096 * 4096, 2359296, 1 << 20, 300001):
    xs = [(torch.randn(n, generator=g) * 0.05).to(dev) for _ in range(max(2, (320 << 20) // (4 * n)))]
    k = n // 2
    want = float(torch.sort(xs[0].abs())[0][k - 1])
    row = []
    for knob in (0, 34):
        L.set_tuning(2, knob)
        got = float(ops.kth_value(xs[0], k, True))
        for i in range(10):
            ops.kth_value(xs[i % len(xs)], k, True)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(100):
            ops.kth_value(xs[i % len(xs)], k, True)
        b.record()
        torch.cuda.synchronize()
        row.append((a.elapsed_time(b) * 10.0, got == want))
        L.set_tuning(2, 0)
    print("n = %9d fp32: one launch %6.1f us (exact: %s)   three launches %6.1f us (exact: %s)   %5.1f MB" % (
        n, row[0][0], row[0][1], row[1][0], row[1][1], n * 4 / 1e6), flush=True)

# ---- the fp32 percentile (two selectors), one tensor and four cached batches
import numpy as np  # noqa: E402


def pct_ref(parts, alpha):
    x = np.concatenate([p.float().cpu().numpy().reshape(-1) for p in parts])
    srt = np.sort(x, kind="stable")
    n = x.size
    neg, pos = int((x < 0).sum()), int((x >= 0).sum())
    kmax = n - max(int(np.rint(pos * alpha)), 0)
    kmin = max(int(np.rint(neg * alpha)), 1)
    return float(srt[kmin - 1]) if neg > 0 else 0.0, float(srt[min(max(kmax, 1), n) - 1]) if pos > 0 else 0.0


for name, shapes in (("one tensor 4096x4096", [(4096, 4096)]), ("4 batches of 64x64x56x56 (config 1's activations)", [(64, 64, 56, 56)] * 4),
                     ("4 batches of 64x197x384 (config 3's, fp32)", [(64, 197, 384)] * 4)):
    sets = [[torch.randn(s, generator=g).to(dev).reshape(1, -1) for s in shapes] for _ in range(3)]
    for alpha in (1e-3, 1e-5):
        want = pct_ref(sets[0], alpha)
        row = []
        for knob in (0, 34):
            L.set_tuning(2, knob)
            mn, mx = ops.percentile_select(sets[0], alpha, per_channel=False)
            ok = (float(mn), float(mx)) == want
            for i in range(5):
                ops.percentile_select(sets[i % 3], alpha, per_channel=False)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(60):
                ops.percentile_select(sets[i % 3], alpha, per_channel=False)
            b.record()
            torch.cuda.synchronize()
            row.append((a.elapsed_time(b) * 1e3 / 60, ok))
            L.set_tuning(2, 0)
        mb = sum(int(np.prod(s)) for s in shapes) * 4 / 1e6
        print("percentile alpha %g, %s (%.0f MB): one launch %6.1f us (exact: %s)   three launches %6.1f us (exact: %s)" % (
            alpha, name, mb, row[0][0], row[0][1], row[1][0], row[1][1]), flush=True)
