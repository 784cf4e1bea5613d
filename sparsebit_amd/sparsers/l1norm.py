"""Unstructured L1 sparser (mirrors sparsebit/sparse/sparsers/l1norm.py:14-26).

The reference sorts all of |w| to read ONE element of the sorted array; here the
threshold is an exact k-th order statistic found by a three-pass radix select
(three streaming reads of w, no sort, no extra copy of the data) and the mask is
one more streaming pass -- or is never materialised at all when the consumer is
the fused mask+QDQ kernel (`calc_threshold` + lsq.Quantizer.forward_masked).
Ties with the threshold are pruned (`>` is strict), exactly like the reference.
Structured pruning is outside the hot path.
"""
import torch

from . import Sparser as BaseSparser
from . import register_sparser
from .. import dist as sbq_dist
from .. import ops
from .. import select


@register_sparser
class Sparser(BaseSparser):
    STRATEGY = "l1norm"

    def __init__(self, config, opr=None):
        super(Sparser, self).__init__(config, opr)

    def calc_threshold(self, x):
        """0-d fp32 device tensor: sort(|x|)[min(int(n*ratio), n-1)]   (l1norm.py:21-24)"""
        data = x.detach().contiguous()
        n = data.numel()
        thresh_idx = min(int(n * self.ratio), n - 1)
        if not sbq_dist.active():
            return ops.kth_value(data, thresh_idx + 1, use_abs=True)  # the three radix passes in one call
        vals = select.kth_values([data], [[thresh_idx + 1]], ops.HipSelectBackend(), True, 0, False, data.device)
        return vals.reshape(())

    def calc_mask(self, x):
        if self.ratio == 0.0:
            return torch.ones_like(x)
        if self.type == "unstructed":
            thresh = self.calc_threshold(x)
            return ops.mask_from_threshold(x.detach(), thresh)
        raise NotImplementedError(
            "only the unstructured L1 masker is on the MI355X hot path (type={})".format(self.type)
        )
