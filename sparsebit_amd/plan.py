"""Launch plans: everything `Quantizer.forward` decides per call, decided ONCE.

The reference quantizes twice per layer per step (sparsebit/quantization/modules/conv.py:37-42 ->
quantizers/base.py:55-64 -> uniform.py:14-16 -> quant_tensor.py:74-80,159-185) and every one of those calls walks
Python: the enable switches, `_qparams_preprocess`, the backend table, the same-device assert, contiguity and geometry
of the operands, the dtype ids, the workspace lookup, fifteen ctypes argument conversions.  Against a 3-11 us kernel
that walk was ~15 us (tools/api_overhead.py) -- the whole cost of a small model's quantized forward
(profiles/r04_reference_qmodel_on_device.log: 1.3 ms for ResNet-20's 64 quantizers whose kernels sum to < 0.2 ms).

A QdqPlan is built the first time a calibrated quantizer meets an input of a given (shape, dtype, device) under
no_grad / without anything asking for a gradient, and holds the foreign function, its pre-converted constant
arguments and the keys it is valid for.  The planned forward is

    y = torch.empty(...)                                  one allocation (the C ABI is caller-allocates)
    fn(x.data_ptr(), ..., y.data_ptr(), ..., stream)      one foreign call

and everything else is a handful of identity / integer comparisons:
  * the quantizer's STRUCTURE version (`_sv`, bumped by Quantizer.__setattr__ whenever scale / zero_point are
    re-bound or a switch -- use_quant, fake_fused, export_onnx, backend -- changes) and its descriptor's version
    (set_bit / set_symmetric): a re-calibration, BN fusion, enable_export_onnx ... all invalidate the plan;
  * the input's shape, dtype, device and contiguity.
scale / zero_point VALUES are read by the kernel from device memory at launch, so an optimizer's in-place update of a
learnable step size needs no invalidation.  The TensorRT backend's `zero_point == 0` assertion
(quant_tensor.py:131-134) is made when the plan is built and again whenever the zero point's in-place version moved.

Anything a plan does not cover (autograd, export mode, exotic quantizers, non-contiguous or empty inputs, qparams
that are not contiguous fp32) takes the generic route, which stays the definition of the result: a planned forward
is bit-identical to it (tests/test_gpu_r05.py).
"""
import torch

from . import lib as L
from .common import Backend

_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = getattr(torch._C, "_cuda_getDevice", None)


# process-wide default of the output dtype (quant_tensor.keep_input_dtype) and an epoch that moves with it and with every
# structural change of ANY quantizer: one integer that a captured graph of a whole model compares (sparsebit_amd.graph)
_keep_default = False
_epoch = 0


def bump_epoch():
    global _epoch
    _epoch += 1


def epoch():
    return _epoch


def set_keep_default(flag):
    global _keep_default
    _keep_default = bool(flag)
    bump_epoch()


def keep_default():
    return _keep_default


_enabled = True


def set_enabled(flag):
    """A/B switch (benchmarks, tests): False sends every forward through the generic route"""
    global _enabled
    _enabled = bool(flag)


def _flat_f32(t):
    """the tensor itself when the kernel can read it in place (fp32, contiguous), else None: a converted COPY would
    freeze the values at plan time"""
    if t.dtype is torch.float32 and t.is_contiguous():
        return t
    return None


class QdqPlan:
    __slots__ = ("sv", "qv", "own", "shape", "dtype", "dev_index", "out_dtype", "fn", "head", "tail", "zp", "zp_version", "trt", "keep",
                 "numel", "lsq", "s_home", "z_home", "s_obj", "z_obj", "s_ptr", "z_ptr")

    def __init__(self, quantizer, x, lsq=False):
        """raises ValueError when this (quantizer, x) is not plannable -- the caller then remembers that and keeps
        the generic route"""
        lib = L.load()
        if not x.is_cuda or x.numel() == 0 or not x.is_contiguous() or x.dtype not in L._DTYPES:
            raise ValueError("input")
        bufs = quantizer.__dict__["_buffers"]
        params = quantizer.__dict__["_parameters"]
        # where the two tensors LIVE (the module's parameter or buffer dict) and which objects they were: the plan holds
        # raw addresses, so it keeps the tensors alive and checks on every call that the module still holds the same
        # objects -- nn.Module._apply (.to() / .cuda() / .half()) and load_state_dict(assign=True) re-bind them without
        # going through __setattr__
        self.s_home = params if "scale" in params else bufs
        self.z_home = params if "zero_point" in params else bufs
        scale = self.s_home["scale"]
        zp = self.z_home["zero_point"]
        self.s_obj, self.z_obj = scale, zp
        if scale.device != x.device or zp.device != x.device:
            raise ValueError("device")
        s, z = _flat_f32(scale.detach()), _flat_f32(zp.detach())
        if s is None or z is None:
            raise ValueError("qparams")
        qdesc = quantizer.qdesc
        qmin, qmax = qdesc.qrange
        per_channel = s.numel() > 1
        from .ops import geometry

        outer, C, inner = geometry(x.shape, qdesc.ch_axis, per_channel)
        if s.numel() != C or z.numel() != C:
            raise ValueError("qparams")
        self.trt = (not lsq) and quantizer.backend == Backend.TENSORRT
        self.zp = zp
        self.zp_version = zp._version
        if self.trt:
            from .quantizers.quant_tensor import _assert_symmetric

            _assert_symmetric(zp)
        self.keep = quantizer._out_keeps_dtype()
        self.out_dtype = x.dtype if self.keep else torch.float32
        xd, yd = L._DTYPES[x.dtype], L._DTYPES[self.out_dtype]
        self.sv = quantizer._sv
        self.qv = qdesc.version
        self.own = quantizer.keep_input_dtype is not None  # (the process default only matters to those who follow it)
        self.shape, self.dtype, self.dev_index = x.shape, x.dtype, x.device.index
        self.numel = x.numel()
        self.lsq = lsq
        sp, zpp = s.data_ptr(), z.data_ptr()
        # (the raw addresses are part of the plan: `scale.data = other` / set_() / resize_() move the storage under the
        # same tensor object, which neither _sv nor the identity test below would notice)
        self.s_ptr, self.z_ptr = scale.data_ptr(), zp.data_ptr()
        # call = fn(x_ptr, *head, y_ptr, *tail, stream): plain Python ints, converted by ctypes' argtypes
        if lsq:
            self.fn = lib.sbq_quant_lsq_forward
            self.head = (xd,)
            self.tail = (yd, None, sp, zpp, outer, C, inner, int(qmin), int(qmax))
        elif per_channel:
            self.fn = lib.sbq_quant_perchannel_forward
            self.head = (xd,)
            self.tail = (yd, None, L.Q_NONE, sp, zpp, outer, C, inner, int(qmin), int(qmax), L.ROUND_HALF_EVEN)
        else:
            self.fn = lib.sbq_quant_pertensor_forward
            self.head = (xd,)
            self.tail = (yd, None, L.Q_NONE, sp, zpp, self.numel, int(qmin), int(qmax), L.ROUND_HALF_EVEN)

    def matches(self, quantizer, x):
        return (self.sv == quantizer._sv and self.qv == quantizer.qdesc.version and x.dtype is self.dtype
                and self.s_home.get("scale") is self.s_obj and self.z_home.get("zero_point") is self.z_obj
                and self.s_obj.data_ptr() == self.s_ptr and self.z_obj.data_ptr() == self.z_ptr and x.shape == self.shape
                and x.device.index == self.dev_index and x.is_contiguous()
                and (self.own or self.keep == _keep_default))

    def __call__(self, x):
        idx = self.dev_index
        if _get_device() != idx:  # (rare: a tensor of another GPU than the current one)
            with torch.cuda.device(idx):
                return self(x)
        if self.trt and self.zp._version != self.zp_version:
            from .quantizers.quant_tensor import _assert_symmetric

            _assert_symmetric(self.zp)
            self.zp_version = self.zp._version
        y = torch.empty(self.shape, dtype=self.out_dtype, device=x.device)
        rc = self.fn(x.data_ptr(), *self.head, y.data_ptr(), *self.tail, _raw_stream(idx))
        if rc:
            L.check(rc)
        return y


class PlanCache:
    """per quantizer: the plans of the last few input signatures seen (a quantizer sits on ONE edge of the graph, so its
    inputs usually share a signature; a shared quantizer -- the reference's QAdd feeds both addends through one -- may see
    two) and the signature last found unplannable"""

    __slots__ = ("plan", "more", "refused")
    KEEP = 4

    def __init__(self):
        self.plan = None   # the most recently used plan
        self.more = []     # up to KEEP - 1 others
        self.refused = None

    # a plan holds a foreign function and raw device addresses: copies of the quantizer (copy.deepcopy of a model for an
    # EMA / a checkpoint through pickle) start without one
    def __deepcopy__(self, memo):
        return PlanCache()

    def __reduce__(self):
        return (PlanCache, ())

    def lookup(self, quantizer, x, lsq=False):
        if not _enabled or _raw_stream is None or _get_device is None:
            return None
        p = self.plan
        if p is not None and p.matches(quantizer, x):
            return p
        for i, q in enumerate(self.more):
            if q.matches(quantizer, x):
                self.more[i], self.plan = p, q
                return q
        key = (quantizer._sv, quantizer.qdesc.version, x.shape, x.dtype, x.device.index, x.is_contiguous())
        if self.refused == key:
            return None
        try:
            q = QdqPlan(quantizer, x, lsq)
        except ValueError:
            self.refused = key
            return None
        if p is not None:
            # (plans of an older structure version can never match again: drop them instead of keeping them alive)
            self.more = [m for m in [p] + self.more if m.sv == q.sv and m.qv == q.qv][: self.KEEP - 1]
        self.plan = q
        return q
