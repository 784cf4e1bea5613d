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
    MAX_SHARDED_CHANNELS = 256  # 8 MB of histograms per pass

    def __init__(self, config, qdesc):
        super(Observer, self).__init__(config, qdesc)
        self.alpha = config.OBSERVER.PERCENTILE.ALPHA

    def calc_minmax(self):
        shards = self._shards()
        self.data_cache.reset()
        x0 = shards[0]
        rows_fast = (
            self.is_perchannel
            and self.ch_axis == 0
            and len(shards) == 1
            and not sbq_dist.active()
            and x0[0].numel() <= L.ROWSEL_MAX
        )
        if rows_fast:
            # a [C, inner] weight whose rows fit on chip: one workgroup per row, one read
            mn, mx = ops.percentile_rows(x0.reshape(x0.shape[0], -1), self.alpha)
            return self._store_minmax(mn, mx)
        mn, mx = self._radix_minmax(shards)
        return self._store_minmax(mn, mx)

    def _radix_minmax(self, shards):
        dev = shards[0].device
        perch = self.is_perchannel
        C = shards[0].shape[self.ch_axis] if perch else 1
        # percentile.py:27-43: the counts of negative / non-negative elements and the two ranks come out of
        # the first radix histogram on the device, three reads of the data in total.  Single process: the whole
        # protocol is one library call; sharded over ranks: pass by pass with an all-reduce in between.
        if not sbq_dist.active():
            fused = ops.percentile_select(shards, self.alpha, self.ch_axis, perch)
            if fused is not None:
                return fused
        elif perch and C > self.MAX_SHARDED_CHANNELS:
            # The sharded protocol all-reduces an int64 [C, 2, 2048] histogram per pass: 32 KB per channel, 134 MB
            # at C = 4096 -- three times.  That is a bandwidth problem on xGMI, not the latency-bound statistic
            # exchange the design budgets for, and no shipped config calibrates per-channel
            # ACTIVATIONS with the percentile observer (weights are replicated and never take this path).
            raise L.SbqError(
                "sharded per-channel percentile over %d channels would all-reduce %d MB of histograms per pass; "
                "calibrate this quantizer per tensor, or outside dist.sharded_calibration() (every rank then sees "
                "all batches)" % (C, C * 2 * L.RADIX_BINS * 8 >> 20))
        vals, counts = select.kth_values(shards, None, ops.HipSelectBackend(), False, self.ch_axis, perch, dev,
                                         percentile_alpha=self.alpha, n_channels=C)
        zero = torch.zeros(C, dtype=torch.float32, device=dev)
        mn = torch.where(counts[0] > 0, vals[:, 0], zero)
        mx = torch.where(counts[1] > 0, vals[:, 1], zero)
        return mn, mx
