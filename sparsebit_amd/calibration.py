"""Device-resident calibration driver.

The reference's CalibrationRunner (sparsebit/quantization/tools/calibration.py:11-160) walks
the fx graph node by node and keeps every intermediate activation on the CPU between nodes
("more time for less cuda memory occupation", :156): each node pays a host->device->host
round trip per batch, and every observer then runs on CPU tensors.  With 288 GB of HBM per
GPU there is no reason to leave the device:

  * one ordinary forward pass per calibration batch, on the GPU, quantizers disabled;
  * forward-pre hooks hand each quantized operator's float inputs straight to its
    `input_quantizer` -- streaming observers (min-max) fold them into running statistics and
    keep nothing; the others cache device tensors (never concatenated);
  * weights are observed in place;
  * `calc_qparams()` for every quantizer at the end; with `sharded=True` each rank sees only
    its share of the batches and the observers all-reduce their statistics (sparsebit_amd.dist).

This is the reference's default protocol (asym=False: every observer sees FLOAT inputs,
calibration.py:66-115) without the fx walk, so it needs no tracing and works on any module
tree whose quantized operators follow the QuantOpr convention: attributes `input_quantizer`
and/or `weight_quantizer` (+ `weight`), e.g. a reference QuantModel after
`sparsebit_amd.plugin.install()`.
"""
import torch

from . import dist as sbq_dist
from .quantizers.base import Quantizer as BaseQuantizer


def _tensors(args):
    for a in args:
        if isinstance(a, torch.Tensor):
            yield a
        elif isinstance(a, (list, tuple)):
            yield from _tensors(a)


def _live(q):
    return q is not None and not getattr(q, "fake_fused", False)


class DeviceCalibrator:
    def __init__(self, model):
        self.model = model
        self.oprs = [
            (name, m)
            for name, m in model.named_modules()
            if _live(getattr(m, "input_quantizer", None)) or _live(getattr(m, "weight_quantizer", None))
        ]
        if not self.oprs:
            raise ValueError("no module with an input_quantizer / weight_quantizer found")

    @staticmethod
    def _feed(quantizer, x):
        x = x.detach()
        obs = quantizer.observer
        # Stream only when nothing between the hook and the observer is customised: quantizers
        # that transform what is observed (DoReFa) or read the cache themselves (LSQ, LSQ+)
        # override one of these two methods and keep the cache protocol.
        cls = type(quantizer)
        plain = cls.update_observer is BaseQuantizer.update_observer and cls.calc_qparams is BaseQuantizer.calc_qparams
        if plain and getattr(obs, "STREAMING", False):
            quantizer.dims = x.dim()
            obs.consume(x)
        else:
            quantizer.update_observer(x)

    def _hook(self, module, args):
        q = module.input_quantizer
        for t in _tensors(args):
            self._feed(q, t)

    @torch.no_grad()
    def calibrate(self, batches, forward=None, sharded=False):
        """batches: iterable of model inputs (this rank's share when sharded=True);
        forward(model, batch) defaults to model(batch) / model(*batch).  Returns {name: (scale, zp)}."""
        saved = []
        for _, m in self.oprs:  # float forward: quantizers off, restored afterwards
            for q in (getattr(m, "input_quantizer", None), getattr(m, "weight_quantizer", None)):
                if q is not None:
                    saved.append((q, q.use_quant))
                    q.disable_quant()
        handles = [
            m.register_forward_pre_hook(self._hook) for _, m in self.oprs if _live(getattr(m, "input_quantizer", None))
        ]
        was_training = self.model.training
        self.model.eval()
        try:
            for batch in batches:
                if forward is not None:
                    forward(self.model, batch)
                elif isinstance(batch, (list, tuple)):
                    self.model(*batch)
                else:
                    self.model(batch)
        finally:
            for h in handles:
                h.remove()
            self.model.train(was_training)
        out = {}

        def finish():
            for name, m in self.oprs:
                iq, wq = getattr(m, "input_quantizer", None), getattr(m, "weight_quantizer", None)
                if _live(iq):
                    out[name + ".input_quantizer"] = iq.calc_qparams()
                if _live(wq):
                    # weights are replicated on every rank: observed locally, no exchange needed, but a
                    # sharded exchange of identical statistics is harmless and keeps one code path
                    wq.update_observer(m.weight)
                    out[name + ".weight_quantizer"] = wq.calc_qparams()

        if sharded:
            with sbq_dist.sharded_calibration():
                # every streaming min-max observer of the model in ONE collective
                obs = [
                    m.input_quantizer.observer
                    for _, m in self.oprs
                    if _live(getattr(m, "input_quantizer", None))
                    and getattr(m.input_quantizer.observer, "pending", lambda: None)() is not None
                ]
                if obs:
                    for o, (lo, hi) in zip(obs, sbq_dist.allreduce_minmax_many([o.pending() for o in obs])):
                        o.resolve(lo, hi)
                finish()
        else:
            finish()
        for q, flag in saved:
            q.use_quant = flag
        return out
