"""percentile observer (mirrors sparsebit/quantization/observers/percentile.py:16-46)."""
import torch

from . import Observer as BaseObserver
from . import register_observer
from .. import dist as sbq_dist
from .. import lib as L
from .. import ops
from .. import select


@register_observer
class Observer(BaseObserver):
    TYPE = "percentile"

    def __init__(self, config, qdesc):
        super(Observer, self).__init__(config, qdesc)
        self.alpha = config.OBSERVER.PERCENTILE.ALPHA

    def calc_minmax(self):
        return sbq_dist.run_lockstep([self.sharded_minmax_steps()])[0]

    def sharded_minmax_steps(self):
        """One process: no exchange at all (the generator returns at once).  Sharded: per tensor the windowed
        protocol (select.windowed_steps: one read of a 16-bit batch list, two SUM all-reduces), per channel the
        fixed-digit passes (three reads, three SUMs of an int64 [C, 2, 2048] histogram -- 32 KB per channel, which at
        C = 4096 is 134 MB per pass: a bandwidth-bound exchange on xGMI, milliseconds instead of microseconds, and
        still exact; no shipped config calibrates per-channel ACTIVATIONS with this observer, and weights are
        replicated and never take this path)."""
        shards = self._shards()
        self.data_cache.reset()
        x0 = shards[0]
        dev = x0.device
        perch = self.is_perchannel
        if not sbq_dist.active():
            if perch and self.ch_axis == 0 and len(shards) == 1 and x0[0].numel() <= L.ROWSEL_MAX:
                # a [C, inner] weight whose rows fit on chip: one workgroup (or wave) per row, one read
                mn, mx = ops.percentile_rows(x0.reshape(x0.shape[0], -1), self.alpha)
                return self._store_minmax(mn, mx)
            # percentile.py:27-43: the counts of negative / non-negative elements and the two ranks come out of
            # the selection itself on the device; the whole protocol is one library call
            fused = ops.percentile_select(shards, self.alpha, self.ch_axis, perch)
            if fused is not None:
                return self._store_minmax(*fused)
        C = x0.shape[self.ch_axis] if perch else 1
        if not perch:  # (whatever this rank's batch count: the exchange sequence must be the same on every rank)
            vals = yield from select.windowed_steps(shards, ops.HipWindowBackend(x0.dtype), dev, use_abs=False,
                                                    percentile_alpha=self.alpha)
            return self._store_minmax(vals[0:1].clone(), vals[1:2].clone())
        vals, counts = yield from select.kth_values_steps(shards, None, ops.HipSelectBackend(), False, self.ch_axis, perch,
                                                          dev, percentile_alpha=self.alpha, n_channels=C)
        zero = torch.zeros(C, dtype=torch.float32, device=dev)
        mn = torch.where(counts[0] > 0, vals[:, 0], zero)
        mx = torch.where(counts[1] > 0, vals[:, 1], zero)
        return self._store_minmax(mn, mx)
