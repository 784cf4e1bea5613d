"""MSE observer (mirrors sparsebit/quantization/observers/mse.py:28-63).

The reference runs 80 x (fake-quant + subtract + square + mean) passes over the
data.  Here x is read from HBM once per shard: a workgroup keeps a chunk in
registers and walks the 80 shrink candidates over it (sbq_mse_accumulate); the
per-candidate squared errors land in a [C, 80] fp64 table that is SUM-all-reduced
across ranks before the strict-less-than, first-wins argmin (sbq_mse_select).
Per channel the scale applies per row, i.e. the reference's CUDA-kernel semantics
(fake_quant_tensor.cu:181-186), not its CPU mis-broadcast.
"""
import torch

from . import Observer as BaseObserver
from . import register_observer
from .. import dist as sbq_dist
from .. import lib as L
from .. import ops


@register_observer
class Observer(BaseObserver):
    TYPE = "mse"

    def __init__(self, config, qdesc):
        super(Observer, self).__init__(config, qdesc)
        self.alpha = config.OBSERVER.PERCENTILE.ALPHA
        self.best_index = None

    def sharded_minmax_steps(self, shards=None):
        if shards is None:
            shards = self._shards()
            self.data_cache.reset()
        mn, mx = self._local_minmax(shards)
        if sbq_dist.active():
            mn, mx = yield ("max", (mn, mx))
        return self._store_minmax(mn, mx)

    def calc_minmax(self, shards=None):
        return sbq_dist.run_lockstep([self.sharded_minmax_steps(shards)])[0]

    def sharded_qparams_steps(self):
        """Two exchanges when sharded: the MAX of the min-max step, then ONE fp64 SUM that carries the [C, 80]
        squared-error table AND this rank's element count (the pick reads the count on the device: no host round
        trip between the collective and the argmin)."""
        shards = self._shards()
        self.data_cache.reset()
        min_val, max_val = yield from self.sharded_minmax_steps(shards)
        dev = shards[0].device
        perch = self.is_perchannel
        C = min_val.numel()
        qmin, qmax = self.qdesc.qrange
        buf = torch.zeros(C * L.MSE_CANDIDATES + 1, dtype=torch.float64, device=dev)
        sse = buf[:-1].view(C, L.MSE_CANDIDATES)
        n_local = 0
        for x in shards:
            ops.mse_accumulate(x, min_val, max_val, qmin, qmax, self.is_symmetric, sse, self.ch_axis, perch)
            n_local += x.numel() // C
        if sbq_dist.active():
            buf[-1] = float(n_local)
            buf = yield ("sum", buf)
            count = buf[-1:]
        else:
            count = n_local
        scale, zero_point, best = ops.mse_select(buf[:-1].view(C, L.MSE_CANDIDATES), count, min_val, max_val, qmin, qmax,
                                                 self.is_symmetric)
        self.best_index = best
        assert len(self.data_cache) == 0, "free data cache after calc_qparams"
        if not perch:
            # the reference returns the candidate's 0-d qparams per tensor (mse.py:57-61)
            scale, zero_point = scale.reshape(()), zero_point.reshape(())
        return scale, zero_point

    def calc_qparams(self):
        return sbq_dist.run_lockstep([self.sharded_qparams_steps()])[0]
