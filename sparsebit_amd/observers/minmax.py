"""min-max observer (behaviour of sparsebit/quantization/observers/minmax.py:14-25).

Besides the reference's cache-then-reduce protocol it can `consume` batches as they arrive:
min/max are order independent, so each batch is folded into a running [C] (or scalar)
statistic on the device and dropped -- calibration then needs no activation storage at all.
"""
import torch

from . import Observer as BaseObserver
from . import register_observer
from .. import dist as sbq_dist
from .. import lib as L
from .. import ops


@register_observer
class Observer(BaseObserver):
    TYPE = "minmax"
    STREAMING = True  # DeviceCalibrator feeds such observers through consume()

    def __init__(self, config, qdesc):
        super().__init__(config, qdesc)
        self._running = None  # (min, max) folded so far by consume()
        self._state = None  # per tensor: the running state of sbq_minmax_accumulate (device, 8 KB)

    def consume(self, x):
        """Fold one batch into the running statistics without caching it.  Per tensor (every shipped activation
        config) that is ONE launch: the batch's workgroups update the observer's running state with integer
        atomicMin / atomicMax (sbq_minmax_accumulate) -- no partials, no fold launch, no torch.minimum / maximum."""
        if not x.is_cuda:
            x = x.to(self.device, non_blocking=True)
        if not self.is_perchannel:
            xd = x.detach()
            L.require_device(xd)  # (no CPU path: fail here like every other observer would)
            if xd.is_contiguous() and xd.data_ptr() % 16 == 0 and xd.numel() > 0:
                if self._state is None:
                    self._state = ops.minmax_state(xd.device)
                ops.minmax_accumulate(xd, self._state)
                return
        lo, hi, _ = ops.channel_stats(x.detach(), self.ch_axis, self.is_perchannel)
        if self._running is not None:
            lo = torch.minimum(self._running[0], lo)  # NaN-propagating, like torch.min over the union
            hi = torch.maximum(self._running[1], hi)
        self._running = (lo, hi)

    def _drain_state(self):
        """the streamed state, if any, joins the (min, max) pair; the observer is back to 'nothing seen'"""
        if self._state is not None:
            lo, hi = ops.minmax_state_read(self._state)
            self._state = None
            if self._running is not None:
                lo, hi = torch.minimum(self._running[0].reshape(1), lo), torch.maximum(self._running[1].reshape(1), hi)
            self._running = (lo, hi)

    def pending(self):
        """The locally folded (min, max), or None -- lets a calibration driver all-reduce the
        statistics of many observers in one collective (dist.allreduce_minmax_many) and hand
        the result back through `resolve`."""
        self._drain_state()
        return self._running

    def resolve(self, lo, hi):
        """Install globally reduced statistics: calc_minmax will then skip its own exchange."""
        self._running = (lo, hi)
        self._resolved = True

    def _local(self):
        """this rank's folded (min, max): what consume() accumulated and / or the cached batches"""
        self._drain_state()
        running, self._running = self._running, None
        if len(self.data_cache):
            shards = self._shards()
            self.data_cache.reset()
            for x in shards:
                lo, hi, _ = ops.channel_stats(x, self.ch_axis, self.is_perchannel)
                running = (lo, hi) if running is None else (torch.minimum(running[0], lo), torch.maximum(running[1], hi))
        assert running is not None, "No data cached!"
        return running

    def sharded_minmax_steps(self):
        resolved = getattr(self, "_resolved", False)
        self._resolved = False
        running = self._local()
        lo, hi = running if resolved else (yield ("max", running))
        return self._store_minmax(lo, hi)

    def calc_minmax(self):
        if sbq_dist.active():
            return sbq_dist.run_lockstep([self.sharded_minmax_steps()])[0]
        self._resolved = False
        return self._store_minmax(*self._local())
