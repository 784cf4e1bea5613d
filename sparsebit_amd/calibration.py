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

This is the reference's calibration protocol (every observer sees FLOAT inputs -- also with
asym=True, see `layerwise_calibration` -- calibration.py:66-129) without the fx walk, so it needs no tracing and works on any module
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

    # ---- the reference's call sequence (quant_model.py:181-199) -----------------------------------
    # qmodel.prepare_calibration(); for batch in loader: qmodel(batch); qmodel.calc_qparams(asym=...)
    def prepare_calibration(self):
        """Quantizers off, model in eval mode, forward-pre hooks on every operator with a live
        input_quantizer: the caller's own calibration forwards now feed the observers on the device."""
        assert not getattr(self, "_handles", None), "prepare_calibration called twice"
        self._saved = []
        for _, m in self.oprs:  # float forward: quantizers off, restored afterwards
            for q in (getattr(m, "input_quantizer", None), getattr(m, "weight_quantizer", None)):
                if q is not None:
                    self._saved.append((q, q.use_quant))
                    q.disable_quant()
        self._handles = [
            m.register_forward_pre_hook(self._hook) for _, m in self.oprs if _live(getattr(m, "input_quantizer", None))
        ]
        self._was_training = self.model.training
        self.model.eval()

    def abort(self):
        """Leave calibration without computing qparams: hooks off, modes and switches restored, caches dropped."""
        for h in getattr(self, "_handles", None) or []:
            h.remove()
        self._handles = None
        self.model.train(self._was_training)
        for q, flag in self._saved:
            q.use_quant = flag
            q.observer.data_cache.reset()
        self._saved = []

    @torch.no_grad()
    def layerwise_calibration(self, device=None, asym=False, w_quant=False, a_quant=False, sharded=False):
        """Finish calibration: qparams of every live quantizer.  -> {name: (scale, zero_point)}.

        `asym` / `w_quant` / `a_quant` are accepted for signature compatibility with
        CalibrationRunner.layerwise_calibration (tools/calibration.py:66-107).  In the reference they do NOT
        change what any observer sees: `run_feature_calibration` reads `self.builder.storage` -- the FLOAT
        activations -- in both modes (:109-123) and weights are observed as they are (:125-129); the quantized
        replay (`qstorage`, :97-103) only feeds AdaRound's reconstruction (:130-143), which is outside this
        path.  So asym=True yields the same scale / zero_point as asym=False for every quantizer here, and
        the replay is not run (tests/test_gpu_plugin.py::test_device_calibrator_equals_reference_calibration_runner
        pins this against goldens from the real CalibrationRunner in both modes).

        AdaRound (a reference-native weight quantizer that stays registered after plugin.install) needs exactly
        that replay: `reconstruct_qlayer` (tools/calibration.py:130-143) is not part of this driver, so a model
        holding one is refused here instead of being left silently un-reconstructed."""
        assert getattr(self, "_handles", None) is not None, "run prepare_calibration first!"
        for name, m in self.oprs:
            wq = getattr(m, "weight_quantizer", None)
            if _live(wq) and str(getattr(wq, "TYPE", "")).lower() == "adaround":
                self.abort()
                raise NotImplementedError(
                    "%s.weight_quantizer is AdaRound: its layer reconstruction (tools/calibration.py:130-143) is not "
                    "part of the device calibrator -- calibrate this model with the reference's CalibrationRunner "
                    "(plugin.install() without calibrate='device' leaves it in place)" % name
                )
        for h in self._handles:
            h.remove()
        self._handles = None
        self.model.train(self._was_training)
        out = {}
        try:
            self._finish(out, sharded)
        finally:
            # also on an exception inside a calc_qparams: switches restored, nothing stale left behind
            for q, flag in self._saved:
                q.use_quant = flag
            self._saved = []
        return out

    def _finish(self, out, sharded):
        if sharded:
            with sbq_dist.sharded_calibration():
                self._finish_inputs_lockstep(out)
        else:
            self._finish_inputs(out)
        # Weights are replicated on every rank: observed and finished OUTSIDE the sharded context.  Inside it
        # their statistics would be all-reduced as if the W replicas were W distinct shards -- harmless for
        # min/max, but ACIQ's sample count, the percentile ranks and LSQ+'s unbiased std would change.
        grouped = self._calibrate_weights_grouped(out)
        for name, m in self.oprs:
            wq = getattr(m, "weight_quantizer", None)
            if _live(wq) and name not in grouped:
                wq.update_observer(m.weight)
                out[name + ".weight_quantizer"] = wq.calc_qparams()

    def _calibrate_weights_grouped(self, out):
        """The weight quantizers whose calibration is nothing but a plain min-max / MSE observer followed by
        calc_qparams (the uniform quantizer of every shipped PTQ config), all at once: two launches for the model's
        min-max observers, four for its MSE observers (ops.GroupCalibration) instead of 3-5 per layer
        (tools/calibration.py:117-135 loops the layers).  Bit-identical to the per-layer path, which takes whatever
        is not eligible.  -> names handled here."""
        from . import ops
        from .observers.minmax import Observer as MinMaxObserver
        from .observers.mse import Observer as MseObserver
        from .registry import impl_type

        pools = {}
        for name, m in self.oprs:
            wq = getattr(m, "weight_quantizer", None)
            if not _live(wq):
                continue
            w = getattr(m, "weight", None)
            cls = impl_type(wq)
            obs_cls = impl_type(wq.observer)
            plain = (cls.calc_qparams is BaseQuantizer.calc_qparams and cls.update_observer is BaseQuantizer.update_observer
                     and obs_cls in (MinMaxObserver, MseObserver) and isinstance(w, torch.Tensor) and w.is_cuda
                     and len(wq.observer.data_cache) == 0 and (not wq.is_perchannel or wq.qdesc.ch_axis == 0)
                     and w.dim() >= 2 and ops.GroupCalibration.supports(w.detach(), wq.is_perchannel))
            if obs_cls is MseObserver and plain:  # the grouped MSE launch folds at most 96 chunks per row
                plain = (w[0].numel() if wq.is_perchannel else w.numel()) <= 96 * 4096
            if plain:
                pools.setdefault((obs_cls is MseObserver, w.dtype), []).append((name, wq, w.detach()))
        done = set()
        for (is_mse, _), members in pools.items():
            if len(members) < 2:
                continue  # a lone tensor gains nothing
            grp = ops.GroupCalibration([(w, wq.qdesc.qrange[0], wq.qdesc.qrange[1], wq.qdesc.is_symmetric, wq.is_perchannel)
                                        for _, wq, w in members])
            if is_mse:
                scale, zp, index = grp.mse_qparams()
            else:
                _, _, scale, zp = grp.minmax_qparams()
            for i, (name, wq, w) in enumerate(members):
                # (clones: the group's flat buffers are reused by its next launch)
                wq.dims = w.dim()
                wq.observer._store_minmax(grp.views["mn"][i].clone(), grp.views["mx"][i].clone())
                if is_mse:
                    wq.observer.best_index = index[i].clone()
                s_i, z_i = scale[i].clone(), zp[i].clone()
                if not wq.is_perchannel:
                    s_i, z_i = s_i.reshape(()), z_i.reshape(())
                out[name + ".weight_quantizer"] = wq._adopt(s_i, z_i)
                done.add(name)
        return done

    def _finish_inputs(self, out):
        for name, m in self.oprs:
            iq = getattr(m, "input_quantizer", None)
            if _live(iq):
                out[name + ".input_quantizer"] = iq.calc_qparams()

    def _finish_inputs_lockstep(self, out):
        """Sharded: the input quantizers whose calc_qparams is the plain observer call advance their observers'
        exchange protocols TOGETHER (dist.run_lockstep): every step of every min-max / MSE / percentile observer of
        the model travels in one flat collective per kind -- per model: 1 MAX (min-max and the MSE observers' first
        step), 1 fp64 SUM (all MSE tables + counts), 1 + rounds int64 SUMs (all percentile samples / window counts;
        rounds = 1 for 16-bit activations, 2 for fp32) and one device-to-host look at the percentile selections'
        done flags -- instead of the reference-shaped loop over quantizers (tools/calibration.py:102-115) costing
        that many exchanges PER quantizer.  Quantizers that own their calibration (LSQ, LSQ+, PACT, DoReFa) and
        observers without a sharded protocol keep their own calls, after the lock-step ones, in module order."""
        from .registry import impl_type

        gens, owners, rest = [], [], []
        for name, m in self.oprs:
            iq = getattr(m, "input_quantizer", None)
            if not _live(iq):
                continue
            cls = impl_type(iq)
            plain = cls.calc_qparams is BaseQuantizer.calc_qparams and cls.update_observer is BaseQuantizer.update_observer
            if plain and getattr(iq.observer, "sharded_minmax_steps", None) is not None:
                gens.append(iq.observer.sharded_qparams_steps())
                owners.append((name, iq))
            else:
                rest.append((name, iq))
        for (name, iq), (scale, zero_point) in zip(owners, sbq_dist.run_lockstep(gens)):
            out[name + ".input_quantizer"] = iq._adopt(scale, zero_point)
        for name, iq in rest:
            out[name + ".input_quantizer"] = iq.calc_qparams()

    # ---- one call -------------------------------------------------------------------------------
    @torch.no_grad()
    def calibrate(self, batches, forward=None, sharded=False, asym=False):
        """batches: iterable of model inputs (this rank's share when sharded=True);
        forward(model, batch) defaults to model(batch) / model(*batch).  Returns {name: (scale, zp)}."""
        self.prepare_calibration()
        try:
            for batch in batches:
                if forward is not None:
                    forward(self.model, batch)
                elif isinstance(batch, (list, tuple)):
                    self.model(*batch)
                else:
                    self.model(batch)
        except BaseException:
            self.abort()
            raise
        return self.layerwise_calibration(asym=asym, sharded=sharded)
