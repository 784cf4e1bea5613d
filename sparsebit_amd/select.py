"""Exact, shardable k-th order statistic: the three-pass radix protocol.

Serves the percentile observer (sparsebit/quantization/observers/percentile.py:
32-43, which calls torch.kthvalue twice per channel) and the unstructured-mask
threshold (sparsebit/sparse/sparsers/l1norm.py:21-24, which sorts everything) on
data that is cached as a LIST of shards -- never concatenated -- and possibly
spread over ranks.  Per pass: every shard adds into a [C][n_sel][2048] int64
histogram, ranks SUM-all-reduce it, every rank advances the same state.  The
result is bit-identical to a single-process kthvalue over the union.

`backend` supplies histogram/advance/finish: ops.HipSelectBackend in the product;
tests drive the same protocol with a numpy stand-in to cover the gloo path.
"""
from . import dist as sbq_dist


def kth_values(shards, ranks, backend, use_abs, ch_axis, per_channel, device, percentile_alpha=None, n_channels=None):
    """shards: list of tensors (same geometry apart from the batch dim);
    ranks: [C][n_sel] 1-indexed global ranks.  -> tensor [C][n_sel] fp32.

    percentile_alpha (with ranks=None, n_channels=C): the percentile observer's two ranks per channel are
    derived on the device from the first histogram (which does not depend on them) after its all-reduce --
    no sign-count pass over the data, no host round trip.  -> (values [C][2], counts [2][C] = neg, pos)."""
    if percentile_alpha is None:
        C, n_sel = len(ranks), len(ranks[0])
        state = backend.new_state(ranks, device)
    else:
        C, n_sel = n_channels, 2
        state = backend.zero_state(C, n_sel, device)
    counts = None
    for p in range(3):
        hist = backend.new_hist(C, n_sel, device)
        for x in shards:
            backend.histogram(x, state, hist, p, n_sel, use_abs, ch_axis, per_channel)
        sbq_dist.allreduce_sum_(hist)
        if p == 0 and percentile_alpha is not None:
            counts = backend.percentile_ranks(hist, state, percentile_alpha, C)
        backend.advance(hist, state, p, n_sel, C)
    vals = backend.finish(state, n_sel, C, use_abs)
    return vals if percentile_alpha is None else (vals, counts)
