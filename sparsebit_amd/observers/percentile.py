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
        neg = torch.zeros(C, dtype=torch.int64, device=dev)
        pos = torch.zeros(C, dtype=torch.int64, device=dev)
        n_local = 0
        for x in shards:
            ops.sign_counts(x, neg, pos, self.ch_axis, perch)
            n_local += x.numel() // C
        counts = torch.stack([neg, pos])
        sbq_dist.allreduce_sum_(counts)
        n = sbq_dist.allreduce_count(n_local)
        neg_l, pos_l = counts[0].tolist(), counts[1].tolist()
        # percentile.py:36-43 -- Python's round (half to even) on pos*alpha / neg*alpha
        ranks = []
        for c in range(C):
            k_max = n - max(round(pos_l[c] * self.alpha), 0)
            k_min = max(round(neg_l[c] * self.alpha), 1)
            ranks.append([min(max(k_min, 1), n), min(max(k_max, 1), n)])
        vals = select.kth_values(shards, ranks, ops.HipSelectBackend(), False, self.ch_axis, perch, dev)
        zero = torch.zeros(C, dtype=torch.float32, device=dev)
        mn = torch.where(counts[0] > 0, vals[:, 0], zero)
        mx = torch.where(counts[1] > 0, vals[:, 1], zero)
        return mn, mx
