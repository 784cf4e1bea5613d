"""Observer base + calibration data cache (mirrors sparsebit/quantization/observers/base.py).

Differences that are the point of the MI355X design:
  * the cache holds DEVICE tensors and is consumed shard by shard -- the reference's
    torch.cat / transpose / flatten copy of all calibration data (base.py:21-36) is
    never made; the kernels index each cached [outer, C, inner] tensor in place;
  * min/max are folded into running device statistics when a batch arrives, so the
    minmax observer needs no cache at all (288 GB of HBM is for activations);
  * statistics, not data, are all-reduced when calibration is sharded over GPUs
    (sparsebit_amd.dist).
"""
import torch
from torch import nn

from .. import dist as sbq_dist
from .. import ops
from ..common import Granularity, QuantTarget


class DataCache(object):
    def __init__(self, qdesc):
        self.qdesc = qdesc
        self._data_cache = []

    def update(self, data):
        self._data_cache.append(data)

    def reset(self):
        self._data_cache = []

    def __len__(self):
        return len(self._data_cache)

    def get_data_cache(self):
        assert len(self._data_cache), "No data cached!"
        return self._data_cache

    def get_batch_size(self):
        if self.qdesc.target == QuantTarget.WEIGHT:
            return None
        return sum([d.shape[self.qdesc.bs_axis] for d in self._data_cache])

    def get_data_for_calibration(self, granularity: Granularity):
        """Compatibility only (third-party observers written against the reference may
        call it): materialises the channel-first / flat copy the reference builds
        (base.py:21-36).  Nothing in sparsebit_amd uses it.  Batches are joined along
        the batch axis, i.e. per-channel means 'over the union of batches' -- the
        reference joins along ch_axis and so yields k*C channels for k batches
        (SURVEY.md 9 Q3), which no shipped config relies on."""
        assert len(self._data_cache), "No data cached!"
        assert granularity in [Granularity.LAYERWISE, Granularity.CHANNELWISE]
        if granularity == Granularity.CHANNELWISE:
            ch = self.qdesc.ch_axis
            if ch == 0:
                assert len(self._data_cache) == 1, "per-channel weights are observed once"
                data = self._data_cache[0]
            else:
                data = torch.cat(self._data_cache, dim=0).transpose(0, ch)
            return data.flatten(1)
        return torch.cat([d.reshape(-1) for d in self._data_cache], axis=0)


class Observer(nn.Module):
    TYPE = "base"

    def __init__(self, config, qdesc):
        nn.Module.__init__(self)  # by name: see quantizers/base.py (plugin.install() mixes in the reference base)
        self.cfg = config
        self.qdesc = qdesc
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.register_buffer("min_val", torch.tensor(float("-inf")).to(self.device))
        self.register_buffer("max_val", torch.tensor(float("inf")).to(self.device))
        self.data_cache = DataCache(qdesc)
        self.backend = None

    # ---- reference interface -------------------------------------------------------
    def calc_qparams(self):
        min_val, max_val = self.calc_minmax()
        scale, zero_point = self.calc_qparams_with_minmax(min_val, max_val)
        return scale, zero_point

    # ---- sharded calibration: the observer as a generator of exchange requests (dist.run_lockstep) --------------
    # Observers that can combine rank-local statistics define `sharded_minmax_steps` (a generator that returns
    # (min_val, max_val) of the union of all ranks' batches); calc_minmax() under dist.sharded_calibration() drives
    # it alone, a calibration driver drives the generators of ALL observers of a model in lock step, so that each
    # step of their protocols is one collective per model (calibration.DeviceCalibrator._finish).
    sharded_minmax_steps = None

    def sharded_qparams_steps(self):
        """-> (scale, zero_point) of the union; generator (see above)"""
        min_val, max_val = yield from self.sharded_minmax_steps()
        return self.calc_qparams_with_minmax(min_val, max_val)

    def calc_qparams_with_minmax(self, min_val, max_val):
        """observers/base.py:63-79 as one device kernel."""
        qmin, qmax = self.qdesc.qrange
        scale, zero_point = ops.qparams_from_minmax(min_val, max_val, qmin, qmax, self.is_symmetric)
        assert len(self.data_cache) == 0, "free data cache after calc_qparams"
        return scale, zero_point

    def calc_minmax(self):
        raise NotImplementedError

    @property
    def is_perchannel(self):
        return self.qdesc.is_perchannel

    @property
    def is_symmetric(self):
        return self.qdesc.is_symmetric

    @property
    def ch_axis(self):
        return self.qdesc.ch_axis

    # ---- shared device helpers ---------------------------------------------------------
    def _shards(self):
        """Cached batches as contiguous device tensors (moved to HBM if a caller cached
        host copies, like the reference's CalibrationRunner does, tools/calibration.py:38)."""
        out = []
        for d in self.data_cache.get_data_cache():
            if not d.is_cuda:
                d = d.to(self.device, non_blocking=True)
            out.append(d.contiguous())
        return out

    def _minmax_over_shards(self, shards):
        """Exact per-channel (or per-tensor) min/max over the union of shards and ranks."""
        return sbq_dist.allreduce_minmax(*self._local_minmax(shards))

    def _local_minmax(self, shards):
        """... over this rank's shards only"""
        mn = mx = None
        for x in shards:
            a, b, _ = ops.channel_stats(x, self.ch_axis, self.is_perchannel)
            # fold shards with NaN-propagating torch.minimum/maximum on [C] vectors
            mn = a if mn is None else torch.minimum(mn, a)
            mx = b if mx is None else torch.maximum(mx, b)
        return mn, mx

    def _store_minmax(self, mn, mx):
        if not self.is_perchannel:
            mn, mx = mn.reshape(()), mx.reshape(())
        self.min_val = mn.to(self.device)
        self.max_val = mx.to(self.device)
        return self.min_val, self.max_val
