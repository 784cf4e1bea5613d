"""All weight quantizers of a model in one launch (forward), for PTQ evaluation and QAT steps.

In the reference every QuantOpr quantizes its own weight inside its own forward
(sparsebit/quantization/modules/conv.py:30-36, linear.py:30-34: `weight_quantizer(self.weight)`,
sparse variants multiply the mask first, sparse/modules/conv.py:39-43): one kernel launch -- and
one trip through Python, the observer-free `Quantizer.forward` and the autograd machinery -- per
layer per step.  On an MI355X those launches are pure latency: a ResNet-50's 54 weights take
~0.8 ms one by one and ~35 us as ONE grid (tools/group_probe.py).

    group = WeightQuantGroup([(module.weight_quantizer, module.weight, mask_or_None), ...])
    wq = group()              # list of fake-quantized weights, same order, autograd-connected

Tensors the grouped kernel cannot take (rows that are not whole 8-element packs, disabled or
exotic quantizers) silently go through their own quantizer, so the result always equals
`[q(w * mask) for q, w, mask in triples]`.
"""
import torch

from . import ops
from .registry import impl_type
from .quantizers.lsq import Quantizer as LSQQuantizer
from .quantizers.uniform import Quantizer as UniformQuantizer


def _kind(q):
    # exact types only: subclasses (LSQ+, PACT, DoReFa ...) transform their inputs / qparams
    if impl_type(q) is UniformQuantizer:
        return "uniform"
    if impl_type(q) is LSQQuantizer:
        return "lsq"
    return None


class _GroupSTE(torch.autograd.Function):
    """forward: the grouped launch; backward: the grouped STE backward (two launches), LSQ's |scale|
    and gradient scaling applied inside the kernels as in lsq.py:13-21,61-76."""

    @staticmethod
    def forward(ctx, group, *tensors):
        n = len(group.members)
        ctx.group = group
        ctx.save_for_backward(*tensors)
        outs = group._launch()
        return tuple(outs) if n > 1 else outs[0]

    @staticmethod
    def backward(ctx, *gouts):
        group = ctx.group
        n = len(group.members)
        need_w = ctx.needs_input_grad[1:1 + n]
        need_s = ctx.needs_input_grad[1 + n:1 + 2 * n]
        # the grouped kernels read the live weights / scales through the device table: unpacking the saved tensors
        # makes autograd's version check fire if one of them was modified in place since the forward
        saved = ctx.saved_tensors
        if all(g is not None for g in gouts):
            gws, gss = group._backward(list(gouts))  # two launches for the whole model
        else:
            gws, gss = group._backward_one_by_one(saved, gouts)
        gws = [g if k else None for g, k in zip(gws, need_w)]
        gss = [g if k else None for g, k in zip(gss, need_s)]
        return (None, *gws, *gss)


class WeightQuantGroup:
    def __init__(self, triples, out_dtype=None):
        self.triples = [(q, w, m) for q, w, m in triples]
        self.out_dtype = out_dtype
        self._partition()

    def _switches(self):
        """what membership depends on and a quantizer may change later (enable_quant / export mode)"""
        return [(bool(q.is_enable), bool(q.export_onnx)) for q, _, _ in self.triples]

    def _partition(self):
        self._seen_switches = self._switches()
        self.members, self.kinds, self.member_idx, self.rest_idx = [], [], [], []
        for i, (q, w, m) in enumerate(self.triples):
            kind = _kind(q)
            per_channel = q.is_perchannel
            ok = (kind is not None and q.is_enable and not q.export_onnx and w.is_cuda and
                  (not per_channel or q.qdesc.ch_axis == 0) and ops.GroupFakeQuant.supports(w, per_channel))
            if ok and m is not None:
                ok = m.shape == w.shape and m.dtype in (torch.bool, torch.uint8) and m.is_contiguous()
            if ok:
                self.members.append((q, w, m)), self.kinds.append(kind), self.member_idx.append(i)
            else:
                self.rest_idx.append(i)
        # a launch takes tensors that all have a mask or none: split, each half is one grid
        self._halves = []
        for masked in (False, True):
            idx = [k for k, (_, _, m) in enumerate(self.members) if (m is not None) == masked]
            if idx:
                self._halves.append({"idx": idx, "gq": None, "ptrs": None})

    # ---- table management ---------------------------------------------------------------------
    def _build(self, half):
        entries, masks, lsq = [], [], []
        for k in half["idx"]:
            q, w, m = self.members[k]
            qmin, qmax = q.qdesc.qrange
            entries.append((w.detach(), q.scale.detach(), q.zero_point.detach().float(), qmin, qmax))
            masks.append(m)
            lsq.append(self.kinds[k] == "lsq")
        masked = masks[0] is not None
        half["gq"] = ops.GroupFakeQuant(entries, out_dtype=self.out_dtype or torch.float32,
                                        masks=masks if masked else None, lsq=lsq, fresh_outputs=True)
        half["ptrs"] = self._live_pointers(half)

    def _live_pointers(self, half):
        """everything baked into the device table: storage addresses and the integer range (set_bit() changes it)"""
        p = []
        for k in half["idx"]:
            q, w, m = self.members[k]
            qmin, qmax = q.qdesc.qrange
            p += [w.data_ptr(), q.scale.data_ptr(), q.zero_point.data_ptr(), 0 if m is None else m.data_ptr(),
                  int(qmin), int(qmax)]
        return p

    def _launch(self):
        outs = [None] * len(self.members)
        for half in self._halves:
            # re-calibration (or .to()) replaces scale / weight storage: rebuild the table then
            if half["gq"] is None or half["ptrs"] != self._live_pointers(half):
                self._build(half)
            for k, y in zip(half["idx"], half["gq"]()):
                outs[k] = y
        return outs

    # ---- backward -----------------------------------------------------------------------------
    def _build_backward(self, half):
        entries, masks, lsq, want, ratios = [], [], [], [], []
        for k in half["idx"]:
            q, w, m = self.members[k]
            qmin, qmax = q.qdesc.qrange
            entries.append((w.detach(), q.scale.detach(), q.zero_point.detach().float(), qmin, qmax))
            masks.append(m)
            is_lsq = self.kinds[k] == "lsq"
            lsq.append(is_lsq)
            want.append(bool(q.scale.requires_grad))
            ratios.append(q._gs_ratio(w) if is_lsq else 1.0)
        half["gb"] = ops.GroupFakeQuantBackward(entries, masks=masks if masks[0] is not None else None, lsq=lsq,
                                                want_gs=want, gs_ratios=ratios)
        half["bptrs"] = self._live_pointers(half) + [int(v) for v in want]

    def _backward(self, gys):
        gws, gss = [None] * len(self.members), [None] * len(self.members)
        for half in self._halves:
            want = [int(self.members[k][0].scale.requires_grad) for k in half["idx"]]
            if half.get("gb") is None or half["bptrs"] != self._live_pointers(half) + want:
                self._build_backward(half)
            gx, gs = half["gb"]([gys[k] for k in half["idx"]])
            for j, k in enumerate(half["idx"]):
                gws[k] = gx[j]
                gss[k] = None if gs[j] is None else gs[j].reshape(self.members[k][0].scale.shape)
        return gws, gss

    def _backward_one_by_one(self, saved, gouts):
        """a layer whose output took no part in the loss hands in None: per-tensor kernels then"""
        n = len(self.members)
        weights, scales = saved[:n], saved[n:]
        gws, gss = [], []
        for i, (q, _, mask) in enumerate(self.members):
            gy = gouts[i]
            if gy is None:
                gws.append(None), gss.append(None)
                continue
            w, s = weights[i], scales[i]
            lsq = self.kinds[i] == "lsq"
            x = w if mask is None else w * mask
            s_eff = s.detach().abs() if lsq else s.detach()
            zp = q.zero_point.detach().float()
            qmin, qmax = q.qdesc.qrange
            if lsq:
                zp = zp.clamp(qmin, qmax)
            need_s = bool(s.requires_grad)
            gx, gs, _ = ops.fake_quant_backward(x.detach(), gy, s_eff, zp, qmin, qmax, 0, need_s, False, gx_dtype=w.dtype)
            if mask is not None:
                gx = gx * mask
            if gs is not None:
                gs = gs.reshape(s.shape)
                if lsq:
                    gs = gs * q._gs_ratio(w) * torch.sign(s.detach())
            gws.append(gx), gss.append(gs)
        return gws, gss

    # ---- zero-edit integration --------------------------------------------------------------
    def attach(self, root):
        """Hook the group into `root.forward`: a forward-pre hook runs the grouped launch and hands every member
        quantizer its result, so the unmodified `self.weight_quantizer(self.weight)` inside each operator
        (modules/conv.py:30-36) returns it instead of launching its own kernel; a forward hook drops the
        references again.  Members whose operator masks the weight first (`weight * w_mask`) see a different
        tensor and simply run their own kernel.  Returns the two hook handles."""

        def before(module, args):
            for (q, w, _), y in zip(self.triples, self()):
                q._pregrouped = (w, y)

        def after(module, args, output):
            for q, _, _ in self.triples:
                q._pregrouped = None

        return root.register_forward_pre_hook(before), root.register_forward_hook(after)

    # ---- forward ------------------------------------------------------------------------------
    def __call__(self):
        if self._switches() != self._seen_switches:  # a quantizer was enabled / disabled / put in export mode
            self._partition()
        result = [None] * len(self.triples)
        for i in self.rest_idx:
            q, w, m = self.triples[i]
            result[i] = q(w if m is None else w * m)
        if self.members:
            ws = [w for _, w, _ in self.members]
            ss = [q.scale for q, _, _ in self.members]
            if torch.is_grad_enabled() and any(t.requires_grad for t in ws + ss):
                outs = _GroupSTE.apply(self, *ws, *ss)
                outs = (outs,) if len(self.members) == 1 else outs
            else:
                outs = self._launch()
            for i, y in zip(self.member_idx, outs):
                result[i] = y
        return result
