"""hipGraph capture of a whole quantized forward: the host leaves the path.

A quantized model's eval-mode forward is a fixed sequence of small kernels -- two quantizer launches per layer
(sparsebit/quantization/modules/conv.py:37-42 -> quantizers/base.py:55-64) between the layer's own kernels -- and on an
MI355X the host, not the GPU, sets its pace: the reference's QuantModel(resnet20) forward at batch 16 took 1.3 ms for
< 0.2 ms of kernels (profiles/r04_reference_qmodel_on_device.log).  Launch plans (sparsebit_amd.plan) shorten every
call; a captured graph removes them:

    fwd = sparsebit_amd.graph.capture(model, example_input)     # warm-up, then ONE hipGraph of the forward
    y = fwd(x)                                                  # copy x into the static input, replay, return outputs

What makes a replay valid, and how it is kept valid:
  * every kernel of this library launches on torch's CURRENT stream with caller-allocated operands, so stream capture
    sees all of them, and their scratch buffers are allocated (per stream) during the warm-up runs on the capture
    stream -- nothing allocates or synchronises inside the capture (the TensorRT backend's `zero_point == 0` check is a
    host read: it is resolved during warm-up, once per zero-point tensor, quant_tensor._assert_symmetric);
  * scale / zero_point VALUES are read from device memory at replay, so in-place updates (an optimizer step) are seen;
  * anything STRUCTURAL -- a re-calibration re-binding scale / zero_point, enable / disable_quant, set_bit,
    export mode, a backend switch, the output-dtype default -- moves the process-wide epoch (sparsebit_amd.plan.epoch):
    a replay compares ONE integer and re-captures (default) or raises (on_stale="raise") when it moved;
  * the model must be in eval mode and its forward free of host-side data dependence (the usual CUDA-graph contract).
Outputs live in the graph's static memory: they are valid until the next replay (clone=True returns copies).

freeze_weights=True additionally takes the WEIGHT quantizers out of the replay: every operator's
`weight_quantizer(weight)` is evaluated once, before the capture, and handed to the operator through the quantizer's
`_pregrouped` slot (the mechanism of group.WeightQuantGroup.attach) while the forward is recorded -- an inference
forward re-quantizes constants on every call otherwise (modules/conv.py:37-42 does).  The fake-quantized weights then
are constants OF THE CAPTURE: an in-place change of a weight or of a weight step size is not seen until the next
capture (structural changes still move the epoch and re-capture).
"""
import torch

from . import plan as sbq_plan


class StaleCapture(RuntimeError):
    pass


def _map(obj, fn):
    if isinstance(obj, torch.Tensor):
        return fn(obj)
    if isinstance(obj, (list, tuple)):
        return type(obj)(_map(o, fn) for o in obj)
    if isinstance(obj, dict):
        return {k: _map(v, fn) for k, v in obj.items()}
    return obj


def _tensors(obj, out):
    if isinstance(obj, torch.Tensor):
        out.append(obj)
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            _tensors(o, out)
    elif isinstance(obj, dict):
        for o in obj.values():
            _tensors(o, out)
    return out


class CapturedForward:
    def __init__(self, model, *example_inputs, warmup=3, on_stale="recapture", clone=False, pool=None, freeze_weights=False):
        if on_stale not in ("recapture", "raise"):
            raise ValueError("on_stale must be 'recapture' or 'raise'")
        if model.training:
            raise RuntimeError("capture() records an inference forward: call model.eval() first")
        self.model, self.warmup, self.on_stale, self.clone, self.pool = model, int(warmup), on_stale, clone, pool
        self.freeze_weights = bool(freeze_weights)
        self.captures = 0
        self._static_in = _map(example_inputs, lambda t: t.detach().clone())
        flat = _tensors(self._static_in, [])
        if not flat or not all(t.is_cuda for t in flat):
            raise RuntimeError("capture() needs device tensors as example inputs")
        self._flat_in = flat
        self._capture()

    def _freeze(self):
        """[(quantizer, weight, fake-quantized weight)] of every operator in the QuantOpr convention"""
        frozen = []
        with torch.no_grad():
            for m in self.model.modules():
                q, w = getattr(m, "weight_quantizer", None), getattr(m, "weight", None)
                if q is None or not isinstance(w, torch.Tensor) or not getattr(q, "is_enable", False) or q.export_onnx:
                    continue
                if not hasattr(q, "_pregrouped") or q._pregrouped is not None:
                    continue  # not one of ours, or already served by a WeightQuantGroup
                frozen.append((q, w, q(w)))
        return frozen

    def _capture(self):
        dev = self._flat_in[0].device
        side = torch.cuda.Stream(device=dev)
        # (the frozen weights are written on the CURRENT stream: the side stream waits after they are enqueued)
        frozen = self._freeze() if self.freeze_weights else []
        side.wait_stream(torch.cuda.current_stream(dev))
        try:
            for q, w, y in frozen:
                q._pregrouped = (w, y)  # (not a structural attribute: the epoch does not move)
            with torch.no_grad(), torch.cuda.stream(side):
                for _ in range(max(self.warmup, 1)):  # plans built, workspaces of THIS stream allocated, host checks resolved
                    self.model(*self._static_in)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            epoch = sbq_plan.epoch()
            g = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(g, pool=self.pool, stream=side):
                out = self.model(*self._static_in)
        finally:
            for q, _, _ in frozen:
                q._pregrouped = None
        self._frozen = frozen  # (the graph reads the fake-quantized weights by address)
        if sbq_plan.epoch() != epoch:
            raise RuntimeError("the model changed its quantizers' structure DURING the forward (the epoch moved): "
                               "such a forward cannot be replayed")
        self.graph, self._static_out, self._epoch = g, out, epoch
        # scratch buffers the captured kernels point into (allocated per stream during the warm-up): the graph holds raw
        # addresses, so this object holds the tensors -- a later, larger request on the same stream key REPLACES the
        # cache entry, and without these references the old buffer would be freed under the graph
        from . import ops

        key = (dev.index, side.cuda_stream)
        self._scratch = [d.get(key) for d in (ops._workspaces, ops._gptq_workspaces, ops._select_workspaces)]
        self._stream = side
        self.captures += 1

    def stale(self):
        return sbq_plan.epoch() != self._epoch

    def __call__(self, *inputs):
        if sbq_plan.epoch() != self._epoch:
            if self.on_stale == "raise":
                raise StaleCapture("a quantizer changed (re-calibration, enable / disable, set_bit, export mode ...) "
                                   "since this forward was captured")
            self._capture()
        new = _tensors(inputs, [])
        if len(new) != len(self._flat_in):
            raise ValueError("expected %d input tensor(s), got %d" % (len(self._flat_in), len(new)))
        for dst, src in zip(self._flat_in, new):
            if dst.shape != src.shape or dst.dtype != src.dtype:
                raise ValueError("input %s %s does not match the captured %s %s" % (tuple(src.shape), src.dtype,
                                                                                    tuple(dst.shape), dst.dtype))
            if src.data_ptr() != dst.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return _map(self._static_out, lambda t: t.clone()) if self.clone else self._static_out


def capture(model, *example_inputs, **kw):
    """-> CapturedForward (see the module docstring).  kw: warmup=3, on_stale="recapture" | "raise", clone=False, pool,
    freeze_weights=False"""
    return CapturedForward(model, *example_inputs, **kw)
