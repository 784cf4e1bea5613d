"""ACIQ observer (mirrors sparsebit/quantization/observers/aciq.py:9-124).

GAUS needs only min/max and the element count; LAPLACE needs b = mean|x - mean(x)|, i.e.
two streaming passes (sum, then absolute deviation), both sbq_channel_moments.  The
closed-form clipping thresholds are a handful of fp32 operations on [C] vectors.
Statistics are all-reduced (MAX / SUM) when calibration is sharded.
"""
import math

import torch

from . import Observer as BaseObserver
from . import register_observer
from .. import dist as sbq_dist
from .. import ops
from ..common import QuantTarget

_ALPHA_GAUS_POS = {1: 1.71, 2: 2.15, 3: 2.55, 4: 2.93, 5: 3.28, 6: 3.61, 7: 3.92, 8: 4.2}
_ALPHA_GAUS = {1: 1.24, 2: 1.71, 3: 2.15, 4: 2.55, 5: 2.93, 6: 3.28, 7: 3.61, 8: 3.92}
_ALPHA_LAPLACE = {0: 1.05, 1: 1.86, 2: 2.83, 3: 3.89, 4: 5.03, 5: 6.2, 6: 7.41, 7: 8.64, 8: 9.89}
_ALPHA_LAPLACE_POS = {0: 1.86, 1: 2.83, 2: 3.89, 3: 5.02, 4: 6.2, 5: 7.41, 6: 8.64, 7: 9.89, 8: 11.16}
_AFFINE = (torch.per_channel_affine, torch.per_tensor_affine)


@register_observer
class Observer(BaseObserver):
    TYPE = "aciq"

    def __init__(self, config, qdesc):
        super(Observer, self).__init__(config, qdesc)
        self.distribution = config.OBSERVER.ACIQ.DISTRIBUTION.lower()
        assert self.distribution in ["gaus", "laplace"], "ACIQ observer only support 'gaus' and 'laplace' mode!"
        self.gaus_const = (0.5 * 0.35) * (1 + (math.pi * math.log(4)) ** 0.5)

    def _counts(self, shards):
        C = shards[0].shape[self.ch_axis] if self.is_perchannel else 1
        per_channel = sum(x.numel() // C for x in shards)
        total = sum(x.numel() for x in shards)
        return C, sbq_dist.allreduce_count(per_channel), sbq_dist.allreduce_count(total)

    def calc_laplace_minmax(self, shards):
        C, n_per_channel, _ = self._counts(shards)
        dev = shards[0].device
        s1 = torch.zeros(C, dtype=torch.float64, device=dev)
        for x in shards:
            ops.channel_moments(x, self.ch_axis, self.is_perchannel, s1, torch.zeros_like(s1))
        sbq_dist.allreduce_sum_(s1)
        mean = (s1 / n_per_channel).float()  # aciq.py:67-72: data.mean(1) / data.mean()
        dev_sum = torch.zeros(C, dtype=torch.float64, device=dev)
        for x in shards:
            ops.channel_absdev(x, mean, self.ch_axis, self.is_perchannel, dev_sum)
        sbq_dist.allreduce_sum_(dev_sum)
        b = (dev_sum / n_per_channel).float()
        if not self.is_perchannel:
            b = b.reshape(())
        gmin, _ = self._minmax_over_shards(shards)
        is_half_range = bool(gmin.min() >= 0)
        half = self.qdesc.scheme in _AFFINE and is_half_range
        alpha = (_ALPHA_LAPLACE_POS if half else _ALPHA_LAPLACE)[self.qdesc.bit]
        return ops.aciq_thresholds(None, None, b, alpha, 0.0, 1.0, half)

    def calc_gaus_minmax(self, shards, batch_size):
        _, _, total = self._counts(shards)
        mn, mx = self._minmax_over_shards(shards)
        if not self.is_perchannel:
            mn, mx = mn.reshape(()), mx.reshape(())
        is_half_range = bool(mn.min() >= 0)
        num_elements = total  # aciq.py:99: numel of ALL cached data, per channel or not
        if self.qdesc.target == QuantTarget.FEATURE:
            num_elements /= batch_size
        half = self.qdesc.scheme in _AFFINE and is_half_range
        alpha = (_ALPHA_GAUS_POS if half else _ALPHA_GAUS)[self.qdesc.bit]
        # std = ((max - min) * gaus_const) / sqrt(2 log n); threshold = alpha * std  (aciq.py:102-113)
        return ops.aciq_thresholds(mn, mx, None, alpha, self.gaus_const, (2 * math.log(num_elements)) ** 0.5, half)

    def calc_minmax(self):
        batch_size = self.data_cache.get_batch_size()
        if batch_size is not None:
            batch_size = sbq_dist.allreduce_count(batch_size)
        shards = self._shards()
        self.data_cache.reset()
        if self.distribution == "laplace":
            min_val, max_val = self.calc_laplace_minmax(shards)
        else:
            min_val, max_val = self.calc_gaus_minmax(shards, batch_size)
        self.min_val = min_val.to(self.device)
        self.max_val = max_val.to(self.device)
        return self.min_val, self.max_val
