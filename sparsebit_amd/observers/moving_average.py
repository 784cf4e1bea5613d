"""moving-average observer (mirrors sparsebit/quantization/observers/moving_average.py:8-34).

Per-sample min/max of a whole batch come from ONE reduction launch (the batch axis plays the
channel role of sbq_channel_stats); the order-dependent EMA recurrence over those N scalars
runs in a one-thread kernel (sbq_ema_minmax) in the reference's fp32 arithmetic.  The EMA
depends on the sample order, so this observer does not shard across ranks (SURVEY.md 8e:
replicas only).
"""
import torch

from . import Observer as BaseObserver
from . import register_observer
from .. import ops
from ..common import QuantTarget


@register_observer
class Observer(BaseObserver):
    TYPE = "moving_average"

    def __init__(self, config, qdesc):
        super(Observer, self).__init__(config, qdesc)
        assert (
            hasattr(config.OBSERVER, "MOVING_AVERAGE") and self.qdesc.target == QuantTarget.FEATURE
        ), "Moving_average observer only support feature observing!"
        self.ema_ratio = config.OBSERVER.MOVING_AVERAGE.EMA_RATIO

    def calc_minmax(self):
        shards = self._shards()
        self.data_cache.reset()
        state = torch.zeros(2, dtype=torch.float32, device=shards[0].device)
        has_state = False
        for batch in shards:
            if self.qdesc.bs_axis > 0:
                batch = batch.transpose(0, self.qdesc.bs_axis).contiguous()
            smin, smax, _ = ops.channel_stats(batch, 0, True)  # one value per sample
            ops.ema_minmax(smin, smax, self.ema_ratio, state, has_state)
            has_state = True
        self.min_val = state[0].clone().to(self.device)
        self.max_val = state[1].clone().to(self.device)
        return self.min_val, self.max_val
