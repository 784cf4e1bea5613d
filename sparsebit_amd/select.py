"""Exact, shardable k-th order statistics over data that is cached as a LIST of shards -- never concatenated --
and possibly spread over ranks.

Serves the percentile observer (sparsebit/quantization/observers/percentile.py:32-43, which calls torch.kthvalue
twice per channel) and the unstructured-mask threshold (sparsebit/sparse/sparsers/l1norm.py:21-24, which sorts
everything).  Two protocols, both bit-identical to a single-process kthvalue over the union:

  windowed_steps     a tensor selected as a WHOLE (C == 1: every shipped activation config).  The ranks' SAMPLE
                     histograms are summed (collective #1), every rank derives the same windows from the sum, sweeps
                     its shards ONCE counting below / inside the windows, the counts are summed (collective #2) and
                     placed: a 16-bit tensor is resolved after that one read, fp32 after a second (sweep, SUM) round;
                     a window that missed its rank costs one more round, never a wrong answer.
  kth_values_steps   per channel: three fixed-digit passes; per pass every shard adds into a [C][n_sel][2048] int64
                     histogram, the ranks SUM it, every rank advances the same state.

Both are written as generators of exchange requests (dist.run_lockstep): a calibration driver advances the
protocols of ALL observers of a model together, so the collectives above are per MODEL, not per quantizer.
`backend` supplies the device steps: ops.HipSelectBackend / ops.HipWindowBackend in the product; the gloo tests
drive the same protocols with numpy stand-ins.
"""
from . import dist as sbq_dist


def kth_values_steps(shards, ranks, backend, use_abs, ch_axis, per_channel, device, percentile_alpha=None, n_channels=None):
    """Generator form of kth_values (same arguments, same result as its return value)."""
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
        hist = yield ("sum", hist)
        if p == 0 and percentile_alpha is not None:
            counts = backend.percentile_ranks(hist, state, percentile_alpha, C)
        backend.advance(hist, state, p, n_sel, C)
    vals = backend.finish(state, n_sel, C, use_abs)
    return vals if percentile_alpha is None else (vals, counts)


def kth_values(shards, ranks, backend, use_abs, ch_axis, per_channel, device, percentile_alpha=None, n_channels=None):
    """shards: list of tensors (same geometry apart from the batch dim);
    ranks: [C][n_sel] 1-indexed global ranks.  -> tensor [C][n_sel] fp32.

    percentile_alpha (with ranks=None, n_channels=C): the percentile observer's two ranks per channel are
    derived on the device from the first histogram (which does not depend on them) after its all-reduce --
    no sign-count pass over the data, no host round trip.  -> (values [C][2], counts [2][C] = neg, pos)."""
    return sbq_dist.run_lockstep([kth_values_steps(shards, ranks, backend, use_abs, ch_axis, per_channel, device,
                                                   percentile_alpha, n_channels)])[0]


MAX_ROUNDS = 8  # a missed window is replaced by "everything below / above it": <= 1 + ceil(32 / 11) more rounds


def windowed_steps(shards, backend, device, use_abs=False, percentile_alpha=None, ranks=None):
    """Whole-tensor selection over this rank's `shards` (possibly none) and every other rank's.

    percentile_alpha: the percentile observer's (min, max) -> returns (values [2] fp32, done).  Otherwise `ranks`:
    one or two explicit 1-based GLOBAL ranks -> returns values [len(ranks)].
    Exchanges: one "sum" for the sample, one per round; after the rounds the data type needs (1 for 16-bit, 2 for
    fp32) a "host" look at the done flags decides about further ones -- identical on every rank, since the state is."""
    pct = percentile_alpha is not None
    n_sel = 2 if pct else len(ranks)
    sample = backend.sample(shards, use_abs, device)
    sample = yield ("sum", sample)
    sel = backend.plan(sample, n_sel, percentile_alpha, ranks, device)
    expected = backend.expected_rounds()
    for r in range(MAX_ROUNDS):
        rec = backend.sweep(sel, shards, use_abs, count_signs=pct and r == 0)
        rec = yield ("sum", rec)
        done = backend.advance(sel, rec)
        if r + 1 >= expected:
            flags = yield ("host", done)
            if all(int(f) != 0 for f in flags[:n_sel]):
                break
    else:
        raise RuntimeError("windowed selection did not converge in %d rounds" % MAX_ROUNDS)
    return backend.values(sel)
