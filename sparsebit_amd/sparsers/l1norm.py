"""L1 sparser (mirrors sparsebit/sparse/sparsers/l1norm.py:14-41).

Unstructured (the hot path): the reference sorts all of |w| to read ONE element of the sorted
array; here the threshold is an exact k-th order statistic found by a three-pass radix select
(three streaming reads of w, no sort, no extra copy of the data) and the mask is one more
streaming pass -- or is never materialised at all when the consumer is the fused mask+QDQ
kernel (`calc_threshold` + lsq.Quantizer.forward_masked).  Ties with the threshold are pruned
(`>` is strict), exactly like the reference.

Structured (l1norm.py:27-41): the `int(C * ratio)` output channels with the smallest sum|w| are
zeroed.  The per-row sums come from the same one-read statistics kernel the observers use
(`sbq_channel_stats`, fp64 accumulation -- the reference sums in fp32, so two rows whose sums
agree to fp32 rounding may swap places; the mask is float like the reference's `ones_like`).

Weights are REPLICATED on every rank of a data-parallel job, so the threshold is always a local
order statistic, also inside `dist.sharded_calibration()`: a histogram all-reduce over W replicas
would describe W*n elements while the rank k comes from the local n (the ratio/W quantile).  A
weight that really is row-sharded over ranks is the caller's explicit choice: `calc_threshold(x,
sharded=True)` takes k from the all-reduced element count and all-reduces the histograms.
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

    def calc_threshold(self, x, sharded=False):
        """0-d fp32 device tensor: sort(|x|)[min(int(n*ratio), n-1)]   (l1norm.py:21-24).
        sharded=True: x is this rank's rows of a weight split over ranks; n and the order statistic are global."""
        data = x.detach().contiguous()
        if sharded and sbq_dist.active():
            n = sbq_dist.allreduce_count(data.numel())
            thresh_idx = min(int(n * self.ratio), n - 1)
            # the windowed protocol over the ranks' shards: one read of a 16-bit weight, two small SUM all-reduces
            vals = sbq_dist.run_lockstep([select.windowed_steps([data.reshape(-1)], ops.HipWindowBackend(data.dtype),
                                                                 data.device, use_abs=True, ranks=[thresh_idx + 1])])[0]
            return vals[0].clone().reshape(())
        n = data.numel()
        thresh_idx = min(int(n * self.ratio), n - 1)
        return ops.kth_value(data, thresh_idx + 1, use_abs=True)  # the three radix passes in one call

    def calc_mask(self, x):
        pre = getattr(self, "_premask", None)
        if pre is not None and pre[0] is x:  # computed model-wide a moment ago (premasks): hand it over once
            self._premask = None
            return pre[1]
        if self.ratio == 0.0:
            return torch.ones_like(x)
        if self.type == "unstructed":
            thresh = self.calc_threshold(x)
            return ops.mask_from_threshold(x.detach(), thresh)
        if self.type == "structed":
            data = x.detach().contiguous()
            _, _, rowsum = ops.channel_stats(data, 0, True, want_min=False, want_max=False, want_abssum=True)
            pruned = int(x.shape[0] * self.ratio)
            keep = torch.ones(x.shape[0], dtype=x.dtype, device=x.device)
            if pruned > 0:
                # stable ascending order like torch.sort on the CPU: among equal sums the lower row index goes first
                keep[torch.sort(rowsum, stable=True).indices[:pruned]] = 0
            return keep.reshape([-1] + [1] * (x.dim() - 1)).expand_as(x).contiguous()
        raise NotImplementedError("sparser type {!r} (the reference knows 'unstructed' and 'structed')".format(self.type))


def calc_masks(pairs):
    """Model-wide form of SparseModel.calc_params (sparse/sparse_model.py:107-113 calls every SparseOpr's calc_mask in
    turn, each with its own torch.sort): the L1 thresholds of ALL unstructured layers come out of one selection launch
    (three for fp32 weights; ops.group_kth_value), then one mask pass per layer.
    pairs: [(sparser, weight), ...] -> [mask, ...]; masks equal `sparser.calc_mask(weight)` element for element.
    Layers of another type or with ratio 0 go through their own calc_mask."""
    masks = [None] * len(pairs)
    todo = [i for i, (sp, w) in enumerate(pairs) if sp.ratio != 0.0 and sp.type == "unstructed" and w.is_cuda]
    by_dtype = {}
    for i in todo:
        by_dtype.setdefault(pairs[i][1].dtype, []).append(i)
    for idxs in by_dtype.values():
        if len(idxs) < 2:
            continue
        ws = [pairs[i][1].detach().contiguous() for i in idxs]
        ks = [min(int(w.numel() * pairs[i][0].ratio), w.numel() - 1) + 1 for i, w in zip(idxs, ws)]
        thr = ops.group_kth_value(ws, ks, use_abs=True)
        for j, i in enumerate(idxs):
            masks[i] = ops.mask_from_threshold(ws[j], thr[j])
    for i, (sp, w) in enumerate(pairs):
        if masks[i] is None:
            masks[i] = sp.calc_mask(w)
    return masks


class premasks:
    """`with premasks(pairs): <the reference's per-layer loop>` -- the zero-edit form of calc_masks: the L1 thresholds of
    all unstructured layers of `pairs` ([(sparser, weight), ...]) come out of ONE selection launch up front, and each
    layer's own `sparser.calc_mask(weight)` inside the block (sparse/modules/conv.py:28-29, called layer by layer from
    sparse/sparse_model.py:107-113) returns its share instead of selecting again.  Identity check on the weight tensor;
    whatever is not picked up is dropped at exit.  Layers the grouped launch does not take (structured, ratio 0, CPU
    weights, a lone tensor of its dtype) are left to their own calc_mask."""

    def __init__(self, pairs):
        self.pairs = [(sp, w) for sp, w in pairs if isinstance(sp, Sparser)]
        self.set = []

    def __enter__(self):
        todo = [i for i, (sp, w) in enumerate(self.pairs) if sp.ratio != 0.0 and sp.type == "unstructed" and w.is_cuda]
        by_dtype = {}
        for i in todo:
            by_dtype.setdefault(self.pairs[i][1].dtype, []).append(i)
        for idxs in by_dtype.values():
            if len(idxs) < 2:
                continue
            ws = [self.pairs[i][1].detach().contiguous() for i in idxs]
            ks = [min(int(w.numel() * self.pairs[i][0].ratio), w.numel() - 1) + 1 for i, w in zip(idxs, ws)]
            thr = ops.group_kth_value(ws, ks, use_abs=True)
            for j, i in enumerate(idxs):
                sp, w = self.pairs[i]
                sp._premask = (w, ops.mask_from_threshold(ws[j], thr[j]))
                self.set.append(sp)
        return self

    def __exit__(self, *exc):
        for sp in self.set:
            sp._premask = None
        return False
