"""Functional layer over the C ABI (include/sbq.h): torch tensors in, torch tensors out.

Every function here launches HIP kernels from libsbq.so on the current torch
stream of the tensor's device.  Operands must already be in HBM; there is no CPU
path (lib.require_device raises).  Outputs are allocated here with torch because
the C ABI is caller-allocates (the reference's native layer allocated with
at::empty_like, fake_quant_tensor.cu:80,211).
"""
import ctypes

import torch

from . import lib as L


def geometry(shape, ch_axis, per_channel):
    """[outer, C, inner] of a contiguous tensor quantized along ch_axis (C == 1: per tensor)."""
    numel = 1
    for s in shape:
        numel *= int(s)
    if not per_channel:
        return 1, 1, numel
    outer = 1
    for s in shape[:ch_axis]:
        outer *= int(s)
    inner = 1
    for s in shape[ch_axis + 1:]:
        inner *= int(s)
    return outer, int(shape[ch_axis]), inner


_workspaces = {}


def _workspace(device, nbytes):
    """Grow-only scratch buffer per (device, stream): reductions write their partials here."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


_gptq_workspaces = {}


def _gptq_workspace(device, nbytes):
    """The mat-vec's workspace starts with arrival counters that must be zero before the first call
    and are left zero by every call (include/sbq.h): it gets its own zero-initialised buffer per
    (device, stream) instead of sharing the general scratch, which other kernels scribble on."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _gptq_workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = L.fresh_workspace(max(int(nbytes), 1 << 20), device)
        _gptq_workspaces[key] = buf
    return buf


_select_workspaces = {}


def _select_workspace(device, nbytes):
    """Workspace of the one-call selections (sbq_percentile_select / sbq_kth_value): like the mat-vec's, it holds
    arrival counters and histogram copies that must be zero before the first call and are left zero by every call
    (include/sbq.h), so it is its own zero-initialised buffer per (device, stream)."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _select_workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = L.fresh_workspace(max(int(nbytes), 1 << 20), device)
        _select_workspaces[key] = buf
    return buf


_DTYPE_IDS = {torch.float32: L.F32, torch.float16: L.F16, torch.bfloat16: L.BF16}


def _f32c(t, device):
    if t.dtype != torch.float32:
        t = t.float()
    return t.reshape(-1).contiguous()


def _check_qparams(scale, zero_point, C):
    if scale.numel() != C or zero_point.numel() != C:
        raise L.SbqError("scale/zero_point must have %d element(s), got %d/%d" % (C, scale.numel(), zero_point.numel()))


# ---------------------------------------------------------------------------------
# forward QDQ
# ---------------------------------------------------------------------------------
def fake_quant(x, scale, zero_point, qmin, qmax, ch_axis=0, out_dtype=None, return_q=None,
               rounding=L.ROUND_HALF_EVEN, mask=None, thresh=None):
    """y = (clamp(round(x/s) + round(zp), qmin, qmax) - round(zp)) * s   [quant_tensor.py:182-184]

    scale.numel() > 1 selects per-channel along ch_axis.  out_dtype: torch.float32
    (the reference's output type) or x.dtype.  return_q: None | torch.int8 | torch.uint8 |
    torch.int32 -> also return the integer tensor; "int4" -> a flat uint8 tensor of x.numel()/2
    bytes holding two levels per byte (export.unpack_int4 undoes it).  mask (torch.bool/uint8) or thresh
    (0-d fp32 tensor) fuse the unstructured-sparsity multiply in front of the QDQ.
    """
    dev = L.require_device(x, scale, zero_point, mask, thresh)
    lib = L.load()
    x = x.contiguous()
    per_channel = scale.numel() > 1
    outer, C, inner = geometry(x.shape, ch_axis, per_channel)
    scale = _f32c(scale, dev)
    zero_point = _f32c(zero_point, dev)
    _check_qparams(scale, zero_point, C)
    out_dtype = out_dtype or torch.float32
    if out_dtype not in (torch.float32, x.dtype):
        raise L.SbqError("out_dtype must be float32 or the input dtype")
    y = torch.empty(x.shape, dtype=out_dtype, device=dev)
    q = None
    q_type = L.Q_NONE
    if return_q is not None:
        if return_q in (torch.int8, torch.uint8):
            q_type = L.Q_I8
        elif return_q == torch.int32:
            q_type = L.Q_I32
        elif return_q == "int4":
            q_type = L.Q_I4
        else:
            raise L.SbqError("return_q must be int8, uint8, int32 or 'int4'")
        if q_type == L.Q_I4:
            if x.numel() % 2:
                raise L.SbqError("packed int4 needs an even number of elements")
            q = torch.empty(x.numel() // 2, dtype=torch.uint8, device=dev)  # flat: byte i holds elements 2i, 2i+1
        else:
            q = torch.empty(x.shape, dtype=return_q, device=dev)
    if x.numel() == 0:
        L.check(2)
    with L.device_guard(dev):
        st = L.stream_ptr(dev)
        if mask is None and thresh is None:
            if per_channel:
                rc = lib.sbq_quant_perchannel_forward(L.ptr(x), L.dtype_id(x), L.ptr(y), L.dtype_id(y), L.ptr(q), q_type,
                                                      L.ptr(scale), L.ptr(zero_point), outer, C, inner,
                                                      int(qmin), int(qmax), rounding, st)
            else:
                rc = lib.sbq_quant_pertensor_forward(L.ptr(x), L.dtype_id(x), L.ptr(y), L.dtype_id(y), L.ptr(q), q_type,
                                                     L.ptr(scale), L.ptr(zero_point), x.numel(),
                                                     int(qmin), int(qmax), rounding, st)
        else:
            if mask is not None:
                if mask.shape != x.shape:
                    raise L.SbqError("mask must have the shape of x")
                mask = mask.contiguous()
                if mask.dtype == torch.bool:
                    mask = mask.view(torch.uint8)
                elif mask.dtype != torch.uint8:
                    raise L.SbqError("mask must be bool or uint8")
            if thresh is not None:
                thresh = _f32c(thresh, dev)
            rc = lib.sbq_mask_quant_forward(L.ptr(x), L.dtype_id(x), L.ptr(y), L.dtype_id(y), L.ptr(q), q_type,
                                            L.ptr(mask), L.ptr(thresh), L.ptr(scale), L.ptr(zero_point),
                                            outer, C, inner, int(qmin), int(qmax), rounding, st)
    L.check(rc)
    return (y, q) if return_q is not None else y


def observe_fake_quant(w, qmin, qmax, symmetric, out_dtype=None):
    """min-max observer + qparams + QDQ of a per-channel weight (channel = dim 0) in ONE read of `w`
    (observers/minmax.py:14-25 -> observers/base.py:63-79 -> quantizers/base.py:55-64).
    -> (y, scale [C], zero_point [C], min [C], max [C]); bit-identical to the three separate steps."""
    dev = L.require_device(w)
    lib = L.load()
    w = w.contiguous()
    C = w.shape[0]
    inner = w.numel() // max(C, 1)
    out_dtype = out_dtype or torch.float32
    if out_dtype not in (torch.float32, w.dtype):
        raise L.SbqError("out_dtype must be float32 or the input dtype")
    y = torch.empty(w.shape, dtype=out_dtype, device=dev)
    stats = torch.empty((4, C), dtype=torch.float32, device=dev)  # scale, zero_point, min, max
    if w.numel() == 0:
        L.check(2)
    with L.device_guard(dev):
        ws = _workspace(dev, lib.sbq_stats_workspace_bytes(1, C, inner))
        rc = lib.sbq_observe_quant_perchannel_forward(L.ptr(w), L.dtype_id(w), L.ptr(y), L.dtype_id(y), L.ptr(stats[0]),
                                                      L.ptr(stats[1]), L.ptr(stats[2]), L.ptr(stats[3]), C, inner,
                                                      int(qmin), int(qmax), 1 if symmetric else 0, L.ptr(ws),
                                                      ws.numel(), L.stream_ptr(dev))
    L.check(rc)
    return y, stats[0], stats[1], stats[2], stats[3]


def quantize_only(x, scale, zero_point, qmin, qmax, ch_axis=0, return_q=torch.int8):
    """QuantizeLinear alone: the integer levels without the dequantized tensor (half the writes).
    return_q as in fake_quant (torch.int8 | torch.uint8 | torch.int32 | "int4")."""
    dev = L.require_device(x, scale, zero_point)
    lib = L.load()
    x = x.contiguous()
    per_channel = scale.numel() > 1
    outer, C, inner = geometry(x.shape, ch_axis, per_channel)
    scale = _f32c(scale, dev)
    zero_point = _f32c(zero_point, dev)
    _check_qparams(scale, zero_point, C)
    if return_q in (torch.int8, torch.uint8):
        q_type, q = L.Q_I8, torch.empty(x.shape, dtype=return_q, device=dev)
    elif return_q == torch.int32:
        q_type, q = L.Q_I32, torch.empty(x.shape, dtype=torch.int32, device=dev)
    elif return_q == "int4":
        if x.numel() % 2:
            raise L.SbqError("packed int4 needs an even number of elements")
        q_type, q = L.Q_I4, torch.empty(x.numel() // 2, dtype=torch.uint8, device=dev)
    else:
        raise L.SbqError("return_q must be int8, uint8, int32 or 'int4'")
    if x.numel() == 0:
        L.check(2)
    with L.device_guard(dev):
        rc = lib.sbq_quant_perchannel_forward(L.ptr(x), L.dtype_id(x), None, L.dtype_id(x), L.ptr(q), q_type,
                                              L.ptr(scale), L.ptr(zero_point), outer, C, inner, int(qmin), int(qmax),
                                              L.ROUND_HALF_EVEN, L.stream_ptr(dev))
    L.check(rc)
    return q


def dequantize_linear(q, scale, zero_point, shape=None, ch_axis=0, signed=True, packed_int4=False, out_dtype=None):
    """DequantizeLinear: (q - round(zp)) * scale -> out_dtype (default fp32).  q: int8 / uint8 / int32 levels
    in the tensor's shape, or the flat packed-int4 bytes (then `shape` is required)."""
    dev = L.require_device(q, scale, zero_point)
    lib = L.load()
    q = q.contiguous()
    if packed_int4:
        if shape is None or q.dtype != torch.uint8:
            raise L.SbqError("packed int4 levels are uint8 bytes and need the tensor's shape")
        shape = torch.Size(shape)
        if q.numel() * 2 != shape.numel():
            raise L.SbqError("packed int4: %d bytes cannot hold %d elements" % (q.numel(), shape.numel()))
        q_type = L.Q_I4
    else:
        shape = q.shape
        if q.dtype in (torch.int8, torch.uint8):
            q_type, signed = L.Q_I8, q.dtype == torch.int8
        elif q.dtype == torch.int32:
            q_type = L.Q_I32
        else:
            raise L.SbqError("levels must be int8, uint8 or int32")
    per_channel = scale.numel() > 1
    outer, C, inner = geometry(shape, ch_axis, per_channel)
    scale = _f32c(scale, dev)
    zero_point = _f32c(zero_point, dev)
    _check_qparams(scale, zero_point, C)
    y = torch.empty(shape, dtype=out_dtype or torch.float32, device=dev)
    if y.numel() == 0:
        L.check(2)
    with L.device_guard(dev):
        rc = lib.sbq_dequantize_linear(L.ptr(q), q_type, int(bool(signed)), L.ptr(y), L.dtype_id(y), L.ptr(scale),
                                       L.ptr(zero_point), outer, C, inner, L.stream_ptr(dev))
    L.check(rc)
    return y


class BatchedFakeQuant:
    """Multi-tensor per-channel QDQ: n same-shape tensors, ONE launch (sbq_quant_perchannel_forward_batched).

    Build once per group of weights (the pointer table lives on the device), call every step:
        bq = BatchedFakeQuant(weights, scales, zero_points, qmin, qmax, ch_axis=0, out_dtype=torch.bfloat16)
        outs = bq()            # list of dequantized tensors, bit-identical to per-tensor fake_quant
    The tensors must stay alive and in place (parameters updated in place by an optimizer do)."""

    def __init__(self, xs, scales, zero_points, qmin, qmax, ch_axis=0, out_dtype=None, outs=None):
        if not (0 < len(xs) <= L.MAX_BATCH) or len(scales) != len(xs) or len(zero_points) != len(xs):
            raise L.SbqError("BatchedFakeQuant: 1..%d tensors with one scale/zero_point each" % L.MAX_BATCH)
        self.dev = L.require_device(*xs, *scales, *zero_points)
        x0 = xs[0]
        for x in xs:
            if x.shape != x0.shape or x.dtype != x0.dtype or not x.is_contiguous():
                raise L.SbqError("BatchedFakeQuant: tensors must share shape/dtype and be contiguous")
        self.outer, self.C, self.inner = geometry(x0.shape, ch_axis, True)
        self.xs = list(xs)
        self.scales = [_f32c(s, self.dev) for s in scales]
        self.zps = [_f32c(z, self.dev) for z in zero_points]
        for s_, z_ in zip(self.scales, self.zps):
            _check_qparams(s_, z_, self.C)
        out_dtype = out_dtype or torch.float32
        if out_dtype not in (torch.float32, x0.dtype):
            raise L.SbqError("out_dtype must be float32 or the input dtype")
        self.outs = list(outs) if outs is not None else [torch.empty(x.shape, dtype=out_dtype, device=self.dev) for x in xs]
        self.qmin, self.qmax = int(qmin), int(qmax)
        rows = [[x.data_ptr(), y.data_ptr(), s_.data_ptr(), z_.data_ptr()]
                for x, y, s_, z_ in zip(self.xs, self.outs, self.scales, self.zps)]
        if any(p % 16 for r in rows for p in r[:2]):
            raise L.SbqError("BatchedFakeQuant: tensors must be 16-byte aligned")
        self.table = torch.tensor(rows, dtype=torch.int64).to(self.dev)
        self.x_dt, self.y_dt = L.dtype_id(x0), L.dtype_id(self.outs[0])

    def __call__(self):
        lib = L.load()
        with L.device_guard(self.dev):
            rc = lib.sbq_quant_perchannel_forward_batched(L.ptr(self.table), len(self.xs), self.x_dt, self.y_dt,
                                                          self.outer, self.C, self.inner, self.qmin, self.qmax,
                                                          L.stream_ptr(self.dev))
        L.check(rc)
        return self.outs


class GroupFakeQuant:
    """Model-wide QDQ: tensors of ANY mix of shapes / integer ranges, ONE launch
    (sbq_group_table_build + sbq_quant_group_forward).

        gq = GroupFakeQuant([(w, scale, zero_point, qmin, qmax), ...], out_dtype=torch.float32,
                            masks=None | [mask, ...], lsq=False | [bool, ...])
        outs = gq()      # list of dequantized tensors, bit-identical to per-tensor fake_quant

    Each tensor is quantized along axis 0 when its scale has more than one element, per tensor
    otherwise.  Build once (the descriptor table lives on the device), call every step; tensors,
    scales and masks must stay alive and in place.  `GroupFakeQuant.supports(w)` tells which
    tensors qualify (contiguous, whole 8-element packs per row, 16-byte aligned); the others
    keep going through fake_quant one by one."""

    @staticmethod
    def supports(x, per_channel=True):
        if not (x.is_cuda and x.is_contiguous() and x.dim() >= 1 and x.numel() > 0) or x.data_ptr() % 16:
            return False
        inner = x.numel() // x.shape[0] if per_channel else x.numel()
        return inner % 8 == 0 and x.numel() < (1 << 27)

    def __init__(self, entries, out_dtype=None, masks=None, lsq=False, outs=None, fresh_outputs=False):
        n = len(entries)
        if n == 0:
            raise L.SbqError("GroupFakeQuant: no tensors")
        xs = [e[0] for e in entries]
        self.dev = L.require_device(*xs, *[e[1] for e in entries], *[e[2] for e in entries],
                                    *(masks if masks is not None else []))
        x0 = xs[0]
        out_dtype = out_dtype or torch.float32
        if out_dtype not in (torch.float32, x0.dtype):
            raise L.SbqError("out_dtype must be float32 or the input dtype")
        if masks is not None and len(masks) != n:
            raise L.SbqError("GroupFakeQuant: one mask per tensor (or none at all)")
        lsq = list(lsq) if isinstance(lsq, (list, tuple)) else [bool(lsq)] * n
        self.xs, self.scales, self.zps, self.masks = [], [], [], []
        # fresh_outputs: every call returns views of ONE newly allocated buffer (what an autograd
        # Function must hand out); the table then holds offsets instead of pointers
        self.fresh = bool(fresh_outputs)
        self.out_dtype = out_dtype
        esz = torch.empty(0, dtype=out_dtype).element_size()
        self.offsets, total = [], 0
        for x in xs:  # rows are whole 8-element packs: every tensor's byte size is a multiple of 16
            self.offsets.append(total)
            total += x.numel() * esz
        self.flat_elems = total // esz
        self.numels = [x.numel() for x in xs]
        if self.fresh:
            self.outs = None
        else:
            self.outs = list(outs) if outs is not None else [torch.empty(x.shape, dtype=out_dtype, device=self.dev) for x in xs]
        items = (L.GroupItem * n)()
        for i, (x, scale, zp, qmin, qmax) in enumerate(entries):
            if x.dtype != x0.dtype or not x.is_contiguous():
                raise L.SbqError("GroupFakeQuant: tensors must share a dtype and be contiguous")
            per_channel = scale.numel() > 1
            C = x.shape[0] if per_channel else 1
            scale, zp = _f32c(scale, self.dev), _f32c(zp, self.dev)
            _check_qparams(scale, zp, C)
            m = None
            if masks is not None:
                m = masks[i]
                if m.shape != x.shape or m.dtype not in (torch.bool, torch.uint8) or not m.is_contiguous():
                    raise L.SbqError("mask must be a contiguous bool / uint8 tensor with the shape of x")
                m = m.view(torch.uint8) if m.dtype == torch.bool else m
            self.xs.append(x), self.scales.append(scale), self.zps.append(zp), self.masks.append(m)
            it = items[i]
            it.x, it.scale, it.zero_point = x.data_ptr(), scale.data_ptr(), zp.data_ptr()
            it.flags = L.GROUP_LSQ if lsq[i] else 0
            if self.fresh:
                it.y = self.offsets[i] or None
                it.flags |= L.GROUP_Y_OFFSET
            else:
                y = self.outs[i]
                if y.shape != x.shape or y.dtype != out_dtype or not y.is_contiguous():
                    raise L.SbqError("GroupFakeQuant: outputs must match the inputs' shapes and be contiguous")
                it.y = y.data_ptr()
            it.mask = m.data_ptr() if m is not None else None
            it.C, it.inner = C, x.numel() // C
            it.qmin, it.qmax = int(qmin), int(qmax)
        lib = L.load()
        n_tiles, need = ctypes.c_uint32(0), ctypes.c_size_t(0)
        L.check(lib.sbq_group_table_build(items, n, None, 0, ctypes.byref(n_tiles), ctypes.byref(need)))
        host = torch.empty(need.value, dtype=torch.uint8)
        L.check(lib.sbq_group_table_build(items, n, host.data_ptr(), need.value, ctypes.byref(n_tiles), None))
        self.table = host.to(self.dev)
        self.n, self.n_tiles = n, n_tiles.value
        self.x_dt, self.y_dt = L.dtype_id(x0), _DTYPE_IDS[out_dtype]
        self.has_mask = int(masks is not None)
        self.shapes = [x.shape for x in xs]

    def pointers(self):
        """what the device table captured -- compare to notice tensors that were re-allocated"""
        return [t.data_ptr() for t in self.xs + self.scales + self.zps + [m for m in self.masks if m is not None]]

    def __call__(self):
        lib = L.load()
        with L.device_guard(self.dev):
            flat = None
            if self.fresh:
                flat = torch.empty(self.flat_elems, dtype=self.out_dtype, device=self.dev)
            rc = lib.sbq_quant_group_forward(L.ptr(self.table), self.n, self.n_tiles, self.x_dt, self.y_dt,
                                             self.has_mask, L.ptr(flat), L.stream_ptr(self.dev))
        L.check(rc)
        if not self.fresh:
            return self.outs
        return [t.view(sh) for t, sh in zip(flat.split_with_sizes(self.numels), self.shapes)]


class GroupCalibration:
    """Model-wide weight calibration: the min-max (and MSE) observers + calc_qparams of a whole list of weights in
    two (four) launches instead of 3-5 per layer (tools/calibration.py:117-135 loops the layers).

    entries: list of (weight, qmin, qmax, symmetric, per_channel).  Weights must stay where they are (pointers are
    captured).  `supports(w)` tells which tensors the grouped launch takes (contiguous, 16-byte aligned, rows of
    whole 8-element packs); the caller calibrates the others one by one.  The min-max results are bit-identical to the
    per-tensor ops (channel_stats + qparams_from_minmax); the MSE search (round 6: a lane per (row, candidate), the pick
    in the same launch) equals mse_accumulate + mse_select except where two candidates' losses tie to fp32 rounding --
    there either may be named (include/sbq.h, section 3)."""

    @staticmethod
    def supports(w, per_channel=True):
        inner = w[0].numel() if per_channel else w.numel()
        return bool(w.is_cuda and w.is_contiguous() and w.data_ptr() % 16 == 0 and inner % 8 == 0 and w.numel() > 0
                    and w.dtype in _DTYPE_IDS)

    def __init__(self, entries):
        lib = L.load()
        if not entries:
            raise L.SbqError("GroupCalibration needs at least one tensor")
        self.dev = L.require_device(*[e[0] for e in entries])
        self.dtype = entries[0][0].dtype
        self.tensors = [e[0] for e in entries]
        n = len(entries)
        items = (L.CalibItem * n)()
        off = 0
        self.slices = []
        for i, (w, qmin, qmax, symmetric, per_channel) in enumerate(entries):
            if w.dtype != self.dtype:
                raise L.SbqError("GroupCalibration: tensors must share a dtype")
            if not self.supports(w, per_channel):
                raise L.SbqError("GroupCalibration: tensor %d is not eligible (see supports())" % i)
            C = w.shape[0] if per_channel else 1
            items[i] = L.CalibItem(w.data_ptr(), C, w.numel() // C, off, int(qmin), int(qmax),
                                   L.CALIB_SYMMETRIC if symmetric else 0, 0)
            self.slices.append(slice(off, off + C))
            off += C
        self.n_values = off
        n_rows = ctypes.c_uint32()
        nbytes = ctypes.c_size_t()
        ws_bytes = ctypes.c_size_t()
        L.check(lib.sbq_calib_table_build(items, n, None, 0, ctypes.byref(n_rows), ctypes.byref(nbytes), ctypes.byref(ws_bytes)))
        self.host_table = (ctypes.c_uint8 * nbytes.value)()
        L.check(lib.sbq_calib_table_build(items, n, self.host_table, nbytes.value, ctypes.byref(n_rows), ctypes.byref(nbytes),
                                          ctypes.byref(ws_bytes)))
        self.table = torch.frombuffer(self.host_table, dtype=torch.uint8).to(self.dev)
        self.ws = torch.empty(max(ws_bytes.value, 16), dtype=torch.uint8, device=self.dev)
        self.mn = torch.empty(off, dtype=torch.float32, device=self.dev)
        self.mx = torch.empty(off, dtype=torch.float32, device=self.dev)
        self.scale = torch.empty(off, dtype=torch.float32, device=self.dev)
        self.zp = torch.empty(off, dtype=torch.float32, device=self.dev)
        self.index = torch.empty(off, dtype=torch.int32, device=self.dev)
        # per-tensor views into the flat result buffers, made once (the launches below only refill the buffers)
        self.views = {name: [getattr(self, name)[sl] for sl in self.slices] for name in ("mn", "mx", "scale", "zp", "index")}

    def launch_minmax(self, want_qparams=True):
        """enqueue the two launches; results land in the flat buffers self.mn / mx (/ scale / zp)"""
        lib = L.load()
        with L.device_guard(self.dev):
            rc = lib.sbq_group_minmax_qparams(L.ptr(self.table), self.host_table, _DTYPE_IDS[self.dtype], L.ptr(self.mn),
                                              L.ptr(self.mx), L.ptr(self.scale) if want_qparams else None,
                                              L.ptr(self.zp) if want_qparams else None, L.ptr(self.ws), self.ws.numel(),
                                              L.stream_ptr(self.dev))
        L.check(rc)

    def launch_mse(self):
        """enqueue the launches of the MSE calibration (two for min-max, one for the search + pick); results in self.scale / zp / index"""
        lib = L.load()
        self.launch_minmax(want_qparams=False)
        with L.device_guard(self.dev):
            rc = lib.sbq_group_mse_qparams(L.ptr(self.table), self.host_table, _DTYPE_IDS[self.dtype], L.ptr(self.mn),
                                           L.ptr(self.mx), L.ptr(self.scale), L.ptr(self.zp), L.ptr(self.index), L.ptr(self.ws),
                                           self.ws.numel(), L.stream_ptr(self.dev))
        L.check(rc)

    def minmax_qparams(self):
        """-> per tensor lists (min, max, scale, zero_point): views into four flat fp32 buffers"""
        self.launch_minmax()
        v = self.views
        return v["mn"], v["mx"], v["scale"], v["zp"]

    def mse_qparams(self):
        """min-max statistics, then the 80-candidate MSE search -> per tensor lists (scale, zero_point, best index)"""
        self.launch_mse()
        v = self.views
        return v["scale"], v["zp"], v["index"]


class GroupFakeQuantBackward:
    """STE backward of a whole group in two launches (sbq_quant_group_backward).

        gb = GroupFakeQuantBackward([(w, scale, zero_point, qmin, qmax), ...], masks=None | [...],
                                    lsq=[bool, ...], want_gs=[bool, ...], gs_ratios=[float, ...])
        gxs, gss = gb(gys)     # lists; gss[i] is None where want_gs[i] is False

    gx_i = mask_i * STE'(mask_i * w_i) * gy_i;  gs_i per channel (flat fp32 [C]), for LSQ items already
    multiplied by gs_ratio and sign(scale).  Outputs are views of two buffers allocated per call."""

    def __init__(self, entries, masks=None, lsq=False, want_gs=False, gs_ratios=None, gx_dtype=None):
        n = len(entries)
        if n == 0:
            raise L.SbqError("GroupFakeQuantBackward: no tensors")
        xs = [e[0] for e in entries]
        self.dev = L.require_device(*xs, *[e[1] for e in entries], *[e[2] for e in entries],
                                    *(masks if masks is not None else []))
        x0 = xs[0]
        self.gx_dtype = gx_dtype or x0.dtype
        if self.gx_dtype not in (torch.float32, x0.dtype):
            raise L.SbqError("gx_dtype must be float32 or the input dtype")
        lsq = list(lsq) if isinstance(lsq, (list, tuple)) else [bool(lsq)] * n
        want_gs = list(want_gs) if isinstance(want_gs, (list, tuple)) else [bool(want_gs)] * n
        gs_ratios = list(gs_ratios) if gs_ratios is not None else [1.0] * n
        esz = torch.empty(0, dtype=self.gx_dtype).element_size()
        self.xs, self.scales, self.zps, self.masks = [], [], [], []
        self.gx_off, self.gs_off, self.shapes, self.Cs = [], [], [x.shape for x in xs], []
        gx_total = gs_total = 0
        items = (L.GroupBwdItem * n)()
        for i, (x, scale, zp, qmin, qmax) in enumerate(entries):
            if x.dtype != x0.dtype or not x.is_contiguous():
                raise L.SbqError("GroupFakeQuantBackward: tensors must share a dtype and be contiguous")
            per_channel = scale.numel() > 1
            C = x.shape[0] if per_channel else 1
            scale, zp = _f32c(scale, self.dev), _f32c(zp, self.dev)
            _check_qparams(scale, zp, C)
            m = None
            if masks is not None:
                m = masks[i]
                if m.shape != x.shape or m.dtype not in (torch.bool, torch.uint8) or not m.is_contiguous():
                    raise L.SbqError("mask must be a contiguous bool / uint8 tensor with the shape of x")
                m = m.view(torch.uint8) if m.dtype == torch.bool else m
            self.xs.append(x), self.scales.append(scale), self.zps.append(zp), self.masks.append(m), self.Cs.append(C)
            self.gx_off.append(gx_total)
            gx_total += x.numel() * esz
            self.gs_off.append(gs_total if want_gs[i] else None)
            if want_gs[i]:
                gs_total += C
            it = items[i]
            it.x, it.scale, it.zero_point = x.data_ptr(), scale.data_ptr(), zp.data_ptr()
            it.mask = m.data_ptr() if m is not None else None
            it.gx_offset, it.gs_offset = self.gx_off[i], self.gs_off[i] or 0
            it.C, it.inner = C, x.numel() // C
            it.qmin, it.qmax = int(qmin), int(qmax)
            it.flags = L.GROUP_LSQ if lsq[i] else 0
            it.want_gs = int(want_gs[i])
            it.gs_ratio = float(gs_ratios[i])
        self.gx_elems, self.gs_floats = gx_total // esz, gs_total
        self.numels = [x.numel() for x in xs]
        self.gs_sizes = [C for C, o in zip(self.Cs, self.gs_off) if o is not None]
        lib = L.load()
        need, wsb = ctypes.c_size_t(0), ctypes.c_size_t(0)
        L.check(lib.sbq_group_bwd_table_build(items, n, None, 0, None, None, ctypes.byref(need), ctypes.byref(wsb)))
        self.host_table = torch.empty(need.value, dtype=torch.uint8)
        L.check(lib.sbq_group_bwd_table_build(items, n, self.host_table.data_ptr(), need.value, None, None, None, None))
        self.table = self.host_table.to(self.dev)
        self.workspace = torch.empty(max(wsb.value, 16), dtype=torch.uint8, device=self.dev)
        self.n = n
        self.x_dt, self.gx_dt = L.dtype_id(x0), _DTYPE_IDS[self.gx_dtype]
        self.has_mask = int(masks is not None)
        self.x_dtype = x0.dtype

    def __call__(self, gys):
        if len(gys) != self.n:
            raise L.SbqError("one output gradient per tensor")
        keep = []
        ptrs = (ctypes.c_void_p * self.n)()
        for i, gy in enumerate(gys):
            if gy.shape != self.shapes[i]:
                raise L.SbqError("gradient %d has the wrong shape" % i)
            if gy.dtype != self.x_dtype or not gy.is_contiguous() or gy.data_ptr() % 16:
                gy = gy.to(self.x_dtype).contiguous()
                if gy.data_ptr() % 16:
                    gy = gy.clone()
            keep.append(gy)
            ptrs[i] = gy.data_ptr()
        lib = L.load()
        with L.device_guard(self.dev):
            gx_flat = torch.empty(self.gx_elems, dtype=self.gx_dtype, device=self.dev)
            gs_flat = torch.empty(max(self.gs_floats, 1), dtype=torch.float32, device=self.dev)
            rc = lib.sbq_quant_group_backward(L.ptr(self.table), self.host_table.data_ptr(), self.n, self.x_dt, self.gx_dt,
                                              self.has_mask, ptrs, L.ptr(gx_flat), L.ptr(gs_flat), L.ptr(self.workspace),
                                              self.workspace.numel(), L.stream_ptr(self.dev))
        L.check(rc)
        gxs = [t.view(sh) for t, sh in zip(gx_flat.split_with_sizes(self.numels), self.shapes)]
        parts = iter(gs_flat[:self.gs_floats].split_with_sizes(self.gs_sizes)) if self.gs_sizes else iter(())
        gss = [None if o is None else next(parts) for o in self.gs_off]
        return gxs, gss


# ---------------------------------------------------------------------------------
# LSQ on the raw parameters: one launch forward, two backward
# ---------------------------------------------------------------------------------
def lsq_fake_quant(x, scale, zero_point, qmin, qmax, ch_axis=0, out_dtype=None, mask=None):
    """LSQ forward with the pre-ops inside the kernel: scale = |scale|, zero_point = clamp(zero_point, qmin, qmax)
    (lsq.py:61-62), then the usual QDQ.  `scale` / `zero_point` are the RAW learnable tensors."""
    dev = L.require_device(x, scale, zero_point, mask)
    lib = L.load()
    x = x.contiguous()
    per_channel = scale.numel() > 1
    outer, C, inner = geometry(x.shape, ch_axis, per_channel)
    scale = _f32c(scale, dev)
    zero_point = _f32c(zero_point, dev)
    _check_qparams(scale, zero_point, C)
    out_dtype = out_dtype or torch.float32
    if out_dtype not in (torch.float32, x.dtype):
        raise L.SbqError("out_dtype must be float32 or the input dtype")
    y = torch.empty(x.shape, dtype=out_dtype, device=dev)
    if mask is not None:
        if mask.shape != x.shape or mask.dtype not in (torch.bool, torch.uint8):
            raise L.SbqError("mask must be a bool / uint8 tensor with the shape of x")
        mask = mask.contiguous()
        mask = mask.view(torch.uint8) if mask.dtype == torch.bool else mask
    if x.numel() == 0:
        L.check(2)
    with L.device_guard(dev):
        rc = lib.sbq_quant_lsq_forward(L.ptr(x), L.dtype_id(x), L.ptr(y), L.dtype_id(y), L.ptr(mask), L.ptr(scale),
                                       L.ptr(zero_point), outer, C, inner, int(qmin), int(qmax), L.stream_ptr(dev))
    L.check(rc)
    return y


def lsq_fake_quant_backward(x, gy, scale, zero_point, qmin, qmax, ch_axis=0, need_gs=True, gs_ratio=1.0, gx_dtype=None):
    """-> (gx, gs | None): STE backward on the raw LSQ parameters; gs (flat fp32 [C]) is already the gradient of
    the RAW step size: sum(gy * dq/ds) * gs_ratio * sign(scale)  (lsq.py:13-21,61-76)."""
    dev = L.require_device(x, gy, scale, zero_point)
    lib = L.load()
    x = x.contiguous()
    gy = gy.contiguous()
    x_dtype = x.dtype
    if gy.dtype != x.dtype:  # see fake_quant_backward: an fp32 upstream gradient is kept, x is widened
        if gy.dtype == torch.float32:
            x = x.float()
        else:
            gy = gy.to(x.dtype)
    per_channel = scale.numel() > 1
    outer, C, inner = geometry(x.shape, ch_axis, per_channel)
    scale = _f32c(scale, dev)
    zero_point = _f32c(zero_point, dev)
    _check_qparams(scale, zero_point, C)
    gx_final = gx_dtype or x_dtype
    # (a widened x computes gx in fp32: the kernels take gx in fp32 or in x's type)
    gx = torch.empty(x.shape, dtype=torch.float32 if x.dtype != x_dtype else gx_final, device=dev)
    gs = torch.empty(C, dtype=torch.float32, device=dev) if need_gs else None
    with L.device_guard(dev):
        ws = _workspace(dev, lib.sbq_backward_workspace_bytes(outer, C, inner)) if need_gs else None
        rc = lib.sbq_quant_lsq_backward(L.ptr(x), L.ptr(gy), L.dtype_id(x), L.ptr(gx), L.dtype_id(gx), L.ptr(gs),
                                        L.ptr(scale), L.ptr(zero_point), outer, C, inner, int(qmin), int(qmax),
                                        float(gs_ratio), L.ptr(ws), ws.numel() if ws is not None else 0, L.stream_ptr(dev))
    L.check(rc)
    if gx.dtype != gx_final:
        gx = gx.to(gx_final)
    return gx, gs


# ---------------------------------------------------------------------------------
# STE backward
# ---------------------------------------------------------------------------------
def fake_quant_backward(x, gy, scale, zero_point, qmin, qmax, ch_axis=0, need_gs=True, need_gzp=True,
                        gx_dtype=None, rounding=L.ROUND_HALF_EVEN):
    """-> (gx, gs | None, gzp | None); gs/gzp are flat fp32 [C]  (fake_quant_tensor.cu:97-132).

    x and the upstream gradient share one element type in the kernel.  When they differ and the gradient is fp32
    (a half-precision activation quantized to the default fp32 output), x is widened -- the reference's route,
    quant_tensor.py:82-103 upcasts x and keeps the fp32 grad_y -- so the scale / zero-point gradients are reduced
    from unrounded upstream gradients; gx comes back in x's own type (or gx_dtype).
    A fractional zero point is rounded half-to-even, as torch.round does in the reference's Python (CPU) path that
    the oracle restates (quant_tensor.py:182-184); the reference's CUDA extension uses std::round (half away from
    zero, fake_quant_tensor.cu:59,111) -- the two differ only for a learned zero point at exactly k + 0.5.
    """
    dev = L.require_device(x, gy, scale, zero_point)
    lib = L.load()
    x = x.contiguous()
    gy = gy.contiguous()
    x_dtype = x.dtype
    if gy.dtype != x.dtype:
        if gy.dtype == torch.float32:
            x = x.float()
        else:
            gy = gy.to(x.dtype)
    per_channel = scale.numel() > 1
    outer, C, inner = geometry(x.shape, ch_axis, per_channel)
    scale = _f32c(scale, dev)
    zero_point = _f32c(zero_point, dev)
    _check_qparams(scale, zero_point, C)
    gx_final = gx_dtype or x_dtype
    # (a widened x computes gx in fp32: the kernels take gx in fp32 or in x's type)
    gx = torch.empty(x.shape, dtype=torch.float32 if x.dtype != x_dtype else gx_final, device=dev)
    gs = torch.empty(C, dtype=torch.float32, device=dev) if need_gs else None
    gzp = torch.empty(C, dtype=torch.float32, device=dev) if need_gzp else None
    if x.numel() == 0:
        L.check(2)
    with L.device_guard(dev):
        nbytes = lib.sbq_backward_workspace_bytes(outer, C, inner)
        ws = _workspace(dev, nbytes)
        rc = lib.sbq_quant_perchannel_backward(L.ptr(x), L.ptr(gy), L.dtype_id(x), L.ptr(gx), L.dtype_id(gx),
                                               L.ptr(gs), L.ptr(gzp), L.ptr(scale), L.ptr(zero_point),
                                               outer, C, inner, int(qmin), int(qmax), rounding,
                                               L.ptr(ws), ws.numel(), L.stream_ptr(dev))
    L.check(rc)
    if gx.dtype != gx_final:
        gx = gx.to(gx_final)
    return gx, gs, gzp


# ---------------------------------------------------------------------------------
# observer reductions
# ---------------------------------------------------------------------------------
def channel_stats(x, ch_axis=0, per_channel=True, want_min=True, want_max=True, want_abssum=False):
    """-> (min [C] fp32 | None, max [C] fp32 | None, abssum [C] fp64 | None) in one read of x."""
    dev = L.require_device(x)
    lib = L.load()
    x = x.contiguous()
    outer, C, inner = geometry(x.shape, ch_axis, per_channel)
    mn = torch.empty(C, dtype=torch.float32, device=dev) if want_min else None
    mx = torch.empty(C, dtype=torch.float32, device=dev) if want_max else None
    ab = torch.empty(C, dtype=torch.float64, device=dev) if want_abssum else None
    if x.numel() == 0:
        L.check(2)
    with L.device_guard(dev):
        ws = _workspace(dev, lib.sbq_stats_workspace_bytes(outer, C, inner))
        rc = lib.sbq_channel_stats(L.ptr(x), L.dtype_id(x), outer, C, inner, L.ptr(mn), L.ptr(mx), L.ptr(ab),
                                   L.ptr(ws), ws.numel(), L.stream_ptr(dev))
    L.check(rc)
    return mn, mx, ab


def channel_moments(x, ch_axis=0, per_channel=True, sum_out=None, sumsq_out=None):
    """sum x / sum x^2 per channel, ADDED into fp64 [C] accumulators (created when None)."""
    dev = L.require_device(x, sum_out, sumsq_out)
    lib = L.load()
    x = x.contiguous()
    outer, C, inner = geometry(x.shape, ch_axis, per_channel)
    if sum_out is None:
        sum_out = torch.zeros(C, dtype=torch.float64, device=dev)
    if sumsq_out is None:
        sumsq_out = torch.zeros(C, dtype=torch.float64, device=dev)
    with L.device_guard(dev):
        ws = _workspace(dev, lib.sbq_stats_workspace_bytes(outer, C, inner))
        rc = lib.sbq_channel_moments(L.ptr(x), L.dtype_id(x), outer, C, inner, None, L.ptr(sum_out), L.ptr(sumsq_out),
                                     None, L.ptr(ws), ws.numel(), L.stream_ptr(dev))
    L.check(rc)
    return sum_out, sumsq_out


def channel_absdev(x, center, ch_axis=0, per_channel=True, out=None):
    """sum |x - center[c]| per channel, ADDED into an fp64 [C] accumulator."""
    dev = L.require_device(x, center, out)
    lib = L.load()
    x = x.contiguous()
    outer, C, inner = geometry(x.shape, ch_axis, per_channel)
    center = _f32c(center, dev)
    if center.numel() != C:
        raise L.SbqError("center must have C elements")
    if out is None:
        out = torch.zeros(C, dtype=torch.float64, device=dev)
    with L.device_guard(dev):
        ws = _workspace(dev, lib.sbq_stats_workspace_bytes(outer, C, inner))
        rc = lib.sbq_channel_moments(L.ptr(x), L.dtype_id(x), outer, C, inner, L.ptr(center), None, None, L.ptr(out),
                                     L.ptr(ws), ws.numel(), L.stream_ptr(dev))
    L.check(rc)
    return out


def aciq_thresholds(min_val, max_val, b, alpha, gaus_const, sqrt_2logn, half_range):
    """-> (min, max) clipping thresholds, aciq.py:65-114, correctly rounded fp32 on device."""
    ref = b if b is not None else min_val
    dev = L.require_device(min_val, max_val, b)
    lib = L.load()
    shape = ref.shape
    mn = None if min_val is None else _f32c(min_val, dev)
    mx = None if max_val is None else _f32c(max_val, dev)
    bb = None if b is None else _f32c(b, dev)
    n = ref.numel()
    lo = torch.empty(n, dtype=torch.float32, device=dev)
    hi = torch.empty(n, dtype=torch.float32, device=dev)
    with L.device_guard(dev):
        rc = lib.sbq_aciq_thresholds(L.ptr(mn), L.ptr(mx), L.ptr(bb), n, float(alpha), float(gaus_const), float(sqrt_2logn),
                                     int(bool(half_range)), L.ptr(lo), L.ptr(hi), L.stream_ptr(dev))
    L.check(rc)
    return lo.reshape(shape), hi.reshape(shape)


def minmax_pack(min_val, max_val):
    """-> fp32 [4C] wire buffer of the min/max exchange (include/sbq.h: sbq_minmax_pack)."""
    dev = L.require_device(min_val, max_val)
    lib = L.load()
    mn, mx = _f32c(min_val, dev), _f32c(max_val, dev)
    buf = torch.empty(4 * mn.numel(), dtype=torch.float32, device=dev)
    with L.device_guard(dev):
        rc = lib.sbq_minmax_pack(L.ptr(mn), L.ptr(mx), mn.numel(), L.ptr(buf), L.stream_ptr(dev))
    L.check(rc)
    return buf


def minmax_unpack(buf, shape):
    dev = L.require_device(buf)
    lib = L.load()
    C = buf.numel() // 4
    mn = torch.empty(C, dtype=torch.float32, device=dev)
    mx = torch.empty(C, dtype=torch.float32, device=dev)
    with L.device_guard(dev):
        rc = lib.sbq_minmax_unpack(L.ptr(buf), C, L.ptr(mn), L.ptr(mx), L.stream_ptr(dev))
    L.check(rc)
    return mn.reshape(shape), mx.reshape(shape)


def ema_minmax(sample_min, sample_max, ratio, state, has_state):
    """state[{min,max}] <- EMA over the samples, in order (moving_average.py:23-31)."""
    dev = L.require_device(sample_min, sample_max, state)
    lib = L.load()
    smin, smax = _f32c(sample_min, dev), _f32c(sample_max, dev)
    import numpy as np

    r = np.float32(ratio)  # the reference multiplies fp32 tensors by Python floats: both factors round to fp32
    om = np.float32(1 - ratio)
    with L.device_guard(dev):
        rc = lib.sbq_ema_minmax(L.ptr(smin), L.ptr(smax), smin.numel(), float(r), float(om), L.ptr(state),
                                int(bool(has_state)), L.stream_ptr(dev))
    L.check(rc)
    return state


def qparams_from_minmax(min_val, max_val, qmin, qmax, symmetric):
    """observers/base.py:63-79 on device -> (scale, zero_point), shaped like min_val."""
    dev = L.require_device(min_val, max_val)
    lib = L.load()
    shape = min_val.shape
    mn = _f32c(min_val, dev)
    mx = _f32c(max_val, dev)
    scale = torch.empty_like(mn)
    zp = torch.empty_like(mn)
    with L.device_guard(dev):
        rc = lib.sbq_qparams_from_minmax(L.ptr(mn), L.ptr(mx), mn.numel(), int(qmin), int(qmax), int(bool(symmetric)),
                                         L.ptr(scale), L.ptr(zp), L.stream_ptr(dev))
    L.check(rc)
    return scale.reshape(shape), zp.reshape(shape)


def lsq_init_scale(abssum, count, qmax):
    dev = L.require_device(abssum)
    lib = L.load()
    ab = abssum.reshape(-1).contiguous()
    if ab.dtype != torch.float64:
        ab = ab.double()
    scale = torch.empty(ab.numel(), dtype=torch.float32, device=dev)
    with L.device_guard(dev):
        rc = lib.sbq_lsq_init_scale(L.ptr(ab), ab.numel(), float(count), int(qmax), L.ptr(scale), L.stream_ptr(dev))
    L.check(rc)
    return scale


def mse_accumulate(x, min_val, max_val, qmin, qmax, symmetric, sse, ch_axis=0, per_channel=True):
    """sse[C][80] (fp64) += sum of squared QDQ error of x for each shrink candidate (mse.py:46-61)."""
    dev = L.require_device(x, min_val, max_val, sse)
    lib = L.load()
    x = x.contiguous()
    outer, C, inner = geometry(x.shape, ch_axis, per_channel)
    mn = _f32c(min_val, dev)
    mx = _f32c(max_val, dev)
    if sse.dtype != torch.float64 or sse.numel() != C * L.MSE_CANDIDATES or not sse.is_contiguous():
        raise L.SbqError("sse must be a contiguous float64 [C, 80] tensor")
    if mn.numel() != C or mx.numel() != C:
        raise L.SbqError("min/max must have C elements")
    with L.device_guard(dev):
        ws = _workspace(dev, lib.sbq_mse_workspace_bytes(outer, C, inner))
        rc = lib.sbq_mse_accumulate(L.ptr(x), L.dtype_id(x), outer, C, inner, L.ptr(mn), L.ptr(mx), int(qmin), int(qmax),
                                    int(bool(symmetric)), L.ptr(sse), L.ptr(ws), ws.numel(), L.stream_ptr(dev))
    L.check(rc)
    return sse


def mse_select(sse, count_per_channel, min_val, max_val, qmin, qmax, symmetric):
    """-> (scale [C], zero_point [C], best_index int32 [C])"""
    dev = L.require_device(sse, min_val, max_val)
    lib = L.load()
    mn = _f32c(min_val, dev)
    mx = _f32c(max_val, dev)
    C = mn.numel()
    scale = torch.empty(C, dtype=torch.float32, device=dev)
    zp = torch.empty(C, dtype=torch.float32, device=dev)
    best = torch.empty(C, dtype=torch.int32, device=dev)
    with L.device_guard(dev):
        if isinstance(count_per_channel, torch.Tensor):  # a device double (sharded calibration): no host read
            cnt = count_per_channel
            if cnt.dtype != torch.float64 or cnt.numel() != 1 or cnt.device != dev:
                raise L.SbqError("mse_select: a tensor count must be one float64 on the data's device")
            rc = lib.sbq_mse_select_devcount(L.ptr(sse), L.ptr(cnt), L.ptr(mn), L.ptr(mx), C, int(qmin), int(qmax),
                                             int(bool(symmetric)), L.ptr(scale), L.ptr(zp), L.ptr(best), L.stream_ptr(dev))
        else:
            rc = lib.sbq_mse_select(L.ptr(sse), float(count_per_channel), L.ptr(mn), L.ptr(mx), C, int(qmin), int(qmax),
                                    int(bool(symmetric)), L.ptr(scale), L.ptr(zp), L.ptr(best), L.stream_ptr(dev))
    L.check(rc)
    return scale, zp, best


# ---------------------------------------------------------------------------------
# order statistics
# ---------------------------------------------------------------------------------
def percentile_rows(x2d, alpha):
    """x2d [C, inner] (inner <= 16384) -> (min [C], max [C]) per percentile.py:16-46."""
    dev = L.require_device(x2d)
    lib = L.load()
    x2d = x2d.contiguous()
    C, inner = x2d.shape
    mn = torch.empty(C, dtype=torch.float32, device=dev)
    mx = torch.empty(C, dtype=torch.float32, device=dev)
    with L.device_guard(dev):
        rc = lib.sbq_percentile_rows(L.ptr(x2d), L.dtype_id(x2d), C, inner, float(alpha), L.ptr(mn), L.ptr(mx),
                                     L.stream_ptr(dev))
    L.check(rc)
    return mn, mx


def sign_counts(x, neg, pos, ch_axis=0, per_channel=True):
    """neg/pos (int64 [C]) += count(x < 0) / count(x >= 0)"""
    dev = L.require_device(x, neg, pos)
    lib = L.load()
    x = x.contiguous()
    outer, C, inner = geometry(x.shape, ch_axis, per_channel)
    with L.device_guard(dev):
        rc = lib.sbq_sign_counts(L.ptr(x), L.dtype_id(x), outer, C, inner, L.ptr(neg), L.ptr(pos), L.stream_ptr(dev))
    L.check(rc)


def radix_histogram(x, state, hist, pass_, n_sel, use_abs, ch_axis=0, per_channel=True):
    dev = L.require_device(x, state, hist)
    lib = L.load()
    x = x.contiguous()
    outer, C, inner = geometry(x.shape, ch_axis, per_channel)
    with L.device_guard(dev):
        rc = lib.sbq_radix_histogram(L.ptr(x), L.dtype_id(x), outer, C, inner, int(bool(use_abs)), int(pass_), int(n_sel),
                                     L.ptr(state), L.ptr(hist), L.stream_ptr(dev))
    L.check(rc)


def radix_advance(hist, state, pass_, n_sel, C):
    dev = L.require_device(hist, state)
    lib = L.load()
    with L.device_guard(dev):
        rc = lib.sbq_radix_advance(L.ptr(hist), int(C), int(pass_), int(n_sel), L.ptr(state), L.stream_ptr(dev))
    L.check(rc)


def radix_finish(state, n_sel, C, use_abs):
    dev = L.require_device(state)
    lib = L.load()
    out = torch.empty((C, n_sel), dtype=torch.float32, device=dev)
    with L.device_guard(dev):
        rc = lib.sbq_radix_finish(L.ptr(state), int(C), int(n_sel), int(bool(use_abs)), L.ptr(out), L.stream_ptr(dev))
    L.check(rc)
    return out


def percentile_select(shards, alpha, ch_axis=0, per_channel=True):
    """percentile observer over a list of cached batches, the whole radix select in ONE call (single process).
    -> (min [C], max [C]) fp32"""
    dev = L.require_device(*shards)
    lib = L.load()
    shards = [x.contiguous() for x in shards]
    x0 = shards[0]
    _, C, inner = geometry(x0.shape, ch_axis, per_channel)
    outers = (ctypes.c_int64 * len(shards))()
    ptrs = (ctypes.c_void_p * len(shards))()
    for i, x in enumerate(shards):
        o, c_, in_ = geometry(x.shape, ch_axis, per_channel)
        if x.dtype != x0.dtype or c_ != C:
            raise L.SbqError("percentile_select: batches must share dtype and channel count")
        if in_ != inner:
            # per tensor a batch is one row of numel elements: batches of different sizes do not share a row
            # length -- the caller uses the stepwise protocol for those
            return None
        outers[i], ptrs[i] = o, x.data_ptr()
    mn = torch.empty(C, dtype=torch.float32, device=dev)
    mx = torch.empty(C, dtype=torch.float32, device=dev)
    ws = _select_workspace(dev, lib.sbq_radix_select_workspace_bytes(C, 2))
    with L.device_guard(dev):
        rc = lib.sbq_percentile_select(ptrs, outers, len(shards), L.dtype_id(x0), C, inner, float(alpha), L.ptr(mn), L.ptr(mx),
                                       L.ptr(ws), ws.numel(), L.stream_ptr(dev))
    L.check(rc)
    return mn, mx


def kth_value(x, k, use_abs=False):
    """1-indexed k-th smallest of x (of |x| with use_abs) as a 0-d fp32 tensor: one call -- and for a 16-bit tensor
    one launch and one read of x"""
    dev = L.require_device(x)
    lib = L.load()
    x = x.contiguous()
    out = torch.empty((), dtype=torch.float32, device=dev)
    ws = _select_workspace(dev, lib.sbq_radix_select_workspace_bytes(1, 1))
    with L.device_guard(dev):
        rc = lib.sbq_kth_value(L.ptr(x), L.dtype_id(x), x.numel(), int(bool(use_abs)), int(k), L.ptr(out), L.ptr(ws),
                               ws.numel(), L.stream_ptr(dev))
    L.check(rc)
    return out


_group_kth_workspaces = {}


def group_kth_value(tensors, ks, use_abs=False):
    """k-th smallest of MANY tensors (of their absolute values with use_abs) in one launch (three for fp32): the L1
    thresholds of a whole model (sparse/sparse_model.py:107-113).  -> flat fp32 tensor [len(tensors)].
    Tensors the grouped launch does not take (unaligned, fewer than 8 elements, not contiguous) go through kth_value."""
    dev = L.require_device(*tensors)
    lib = L.load()
    n = len(tensors)
    if n == 0 or n != len(ks):
        raise L.SbqError("group_kth_value: one rank per tensor")
    out = torch.empty(n, dtype=torch.float32, device=dev)
    dtype = tensors[0].dtype
    ok = [t.is_contiguous() and t.data_ptr() % 16 == 0 and t.numel() >= 8 and t.dtype == dtype for t in tensors]
    idx = [i for i in range(n) if ok[i]]
    if idx:
        items = (L.KthItem * len(idx))()
        for j, i in enumerate(idx):
            items[j] = L.KthItem(tensors[i].data_ptr(), tensors[i].numel(), int(ks[i]))
        nbytes = lib.sbq_group_kth_workspace_bytes(len(idx))
        key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
        ws = _group_kth_workspaces.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = L.fresh_workspace(nbytes, dev)
            _group_kth_workspaces[key] = ws
        vals = out if len(idx) == n else torch.empty(len(idx), dtype=torch.float32, device=dev)
        with L.device_guard(dev):
            rc = lib.sbq_group_kth_value(items, len(idx), L.dtype_id(tensors[idx[0]]), int(bool(use_abs)), L.ptr(vals), L.ptr(ws),
                                         ws.numel(), L.stream_ptr(dev))
        L.check(rc)
        if vals is not out:
            out[torch.tensor(idx, device=dev)] = vals
    for i in range(n):
        if not ok[i]:
            out[i] = kth_value(tensors[i], ks[i], use_abs)
    return out


class HipSelectBackend:
    """The three primitives of the sharded exact selection protocol (select.py), on HIP."""

    def new_state(self, ranks, device):
        # ranks: python nested list [C][n_sel] of 1-indexed ranks
        C, n_sel = len(ranks), len(ranks[0])
        st = torch.zeros((C, n_sel, 2), dtype=torch.int64)
        st[:, :, 1] = torch.tensor(ranks, dtype=torch.int64)
        return st.to(device)

    def new_hist(self, C, n_sel, device):
        return torch.zeros((C, n_sel, L.RADIX_BINS), dtype=torch.int64, device=device)

    def zero_state(self, C, n_sel, device):
        return torch.zeros((C, n_sel, 2), dtype=torch.int64, device=device)

    def percentile_ranks(self, hist, state, alpha, C):
        """ranks of the percentile observer from the (all-reduced) pass-0 histogram -> counts [2][C]"""
        dev = L.require_device(hist, state)
        counts = torch.empty((2, C), dtype=torch.int64, device=dev)
        with L.device_guard(dev):
            rc = L.load().sbq_percentile_ranks(L.ptr(hist), C, 2, float(alpha), L.ptr(state), L.ptr(counts),
                                               L.stream_ptr(dev))
        L.check(rc)
        return counts

    def histogram(self, x, state, hist, pass_, n_sel, use_abs, ch_axis, per_channel):
        radix_histogram(x, state, hist, pass_, n_sel, use_abs, ch_axis, per_channel)

    def advance(self, hist, state, pass_, n_sel, C):
        radix_advance(hist, state, pass_, n_sel, C)

    def finish(self, state, n_sel, C, use_abs):
        return radix_finish(state, n_sel, C, use_abs)


# ---------------------------------------------------------------------------------
# streaming per-tensor min-max (one launch per calibration batch)
# ---------------------------------------------------------------------------------
def minmax_state(device):
    """a fresh running state of the streaming per-tensor min-max observer (include/sbq.h: sbq_minmax_accumulate)"""
    st = torch.empty(2048, dtype=torch.int32, device=device)  # SBQ_MINMAX_STATE_WORDS
    with L.device_guard(device):
        L.check(L.load().sbq_minmax_state_reset(L.ptr(st), L.stream_ptr(device)))
    return st


def minmax_accumulate(x, state):
    """fold the whole tensor x into `state`: ONE launch, no fold, no temporary.  -> False when x is not eligible
    (not contiguous / not 16-byte aligned): the caller then takes channel_stats"""
    dev = L.require_device(x, state)
    if not x.is_contiguous() or x.data_ptr() % 16 or x.numel() == 0:
        return False
    with L.device_guard(dev):
        L.check(L.load().sbq_minmax_accumulate(L.ptr(x), L.dtype_id(x), x.numel(), L.ptr(state), L.stream_ptr(dev)))
    return True


def minmax_state_read(state):
    """-> (min, max) as fp32 tensors of shape [1]"""
    dev = L.require_device(state)
    out = torch.empty(2, dtype=torch.float32, device=dev)
    with L.device_guard(dev):
        L.check(L.load().sbq_minmax_state_read(L.ptr(state), L.ptr(out[0:1]), L.ptr(out[1:2]), L.stream_ptr(dev)))
    return out[0:1], out[1:2]


class HipWindowBackend:
    """The device steps of select.windowed_steps (include/sbq.h section 4b): sample / plan / sweep / advance of ONE
    whole-tensor selection over this rank's shards.  All shards share a dtype; a rank may hold none (dtype given)."""

    def __init__(self, dtype):
        self.dtype = dtype
        try:
            self.dtype_id = _DTYPE_IDS[dtype]
        except KeyError:
            raise L.SbqError("libsbq: Kernel Failure, Invalid dtype of Input tensor: %s" % (dtype,))

    def expected_rounds(self):
        return 2 if self.dtype == torch.float32 else 1  # a sweep resolves 11 key bits below the sample's window

    def _chunks(self):
        """this rank's shards in library calls of <= SBQ_MAX_BATCH: the sample histogram and a round's record are SUMS
        over shards (that is what lets the ranks all-reduce them), so a rank with more cached batches than one call
        takes adds the calls' outputs -- every rank runs the SAME exchange sequence whatever its batch count"""
        sh = self.shards
        return [sh[i:i + L.MAX_BATCH] for i in range(0, len(sh), L.MAX_BATCH)] or [[]]

    def _tables(self, shards):
        n = len(shards)
        ptrs = (ctypes.c_void_p * max(n, 1))()
        counts = (ctypes.c_int64 * max(n, 1))()
        for i, x in enumerate(shards):
            if x.dtype != self.dtype:
                raise L.SbqError("windowed selection: every shard must be %s" % self.dtype)
            ptrs[i], counts[i] = x.data_ptr(), x.numel()
        return ptrs, counts, n

    def sample(self, shards, use_abs, device):
        dev = L.require_device(*shards) if shards else device
        self.shards = [x.contiguous() for x in shards]
        total = None
        for part in self._chunks():
            out = torch.empty(L.DIST_SAMPLE_WORDS, dtype=torch.int64, device=dev)
            ptrs, counts, n = self._tables(part)
            with L.device_guard(dev):
                rc = L.load().sbq_dist_select_sample(ptrs, counts, n, self.dtype_id, int(bool(use_abs)), L.ptr(out), L.stream_ptr(dev))
            L.check(rc)
            total = out if total is None else total.add_(out)
        return total

    def plan(self, sample, n_sel, percentile_alpha, ranks, device):
        lib = L.load()
        dev = sample.device
        ws = torch.empty(lib.sbq_dist_select_workspace_bytes(), dtype=torch.uint8, device=dev)  # the plan clears it
        pct = percentile_alpha is not None
        k0 = k1 = 0
        if not pct:
            k0 = int(ranks[0])
            k1 = int(ranks[1]) if n_sel == 2 else 0
        alpha = float(percentile_alpha) if pct else 0.0
        with L.device_guard(dev):
            rc = lib.sbq_dist_select_plan(L.ptr(sample), self.dtype_id, n_sel, int(pct), alpha, k0, k1, L.ptr(ws), ws.numel(),
                                          L.stream_ptr(dev))
        L.check(rc)
        return {"ws": ws, "n_sel": n_sel, "pct": pct, "alpha": alpha, "dev": dev,
                "out": torch.zeros(2, dtype=torch.float32, device=dev), "done": torch.zeros(2, dtype=torch.int32, device=dev)}

    def sweep(self, sel, shards, use_abs, count_signs):
        dev = sel["dev"]
        total = None
        for part in self._chunks():
            rec = torch.empty(L.DIST_ROUND_WORDS, dtype=torch.int64, device=dev)
            ptrs, counts, n = self._tables(part)
            with L.device_guard(dev):
                rc = L.load().sbq_dist_select_sweep(ptrs, counts, n, self.dtype_id, int(bool(use_abs)), sel["n_sel"], int(bool(count_signs)),
                                                    L.ptr(sel["ws"]), sel["ws"].numel(), L.ptr(rec), L.stream_ptr(dev))
            L.check(rc)
            total = rec if total is None else total.add_(rec)
        return total

    def advance(self, sel, rec):
        dev = sel["dev"]
        out = sel["out"]
        with L.device_guard(dev):
            rc = L.load().sbq_dist_select_advance(L.ptr(rec), self.dtype_id, sel["n_sel"], int(sel["pct"]), sel["alpha"],
                                                  L.ptr(sel["ws"]), sel["ws"].numel(), L.ptr(out[0:1]),
                                                  L.ptr(out[1:2]) if sel["pct"] else None, L.ptr(sel["done"]), L.stream_ptr(dev))
        L.check(rc)
        return sel["done"]

    def values(self, sel):
        return sel["out"][: sel["n_sel"]]


def mask_from_threshold(x, thresh):
    """mask = |x| > thresh as torch.bool (l1norm.py:24-25)"""
    dev = L.require_device(x, thresh)
    lib = L.load()
    x = x.contiguous()
    thresh = _f32c(thresh, dev)
    mask = torch.empty(x.shape, dtype=torch.uint8, device=dev)
    with L.device_guard(dev):
        rc = lib.sbq_mask_from_threshold(L.ptr(x), L.dtype_id(x), x.numel(), L.ptr(thresh), L.ptr(mask), L.stream_ptr(dev))
    L.check(rc)
    return mask.view(torch.bool)


# ---------------------------------------------------------------------------------
# GPTQ 4- / 3- / 2-bit mat-vec
# ---------------------------------------------------------------------------------
def vecquantmatmul(bits, x, qweight, out, scales, zeros, group_size=0):
    """out[b,n] += sum_k (scales[n,g]*lvl - zeros[n,g]) * x[b,k], in place (cuda_kernel.cpp:6-62).
    group_size 0 is the reference's un-grouped vecquant{bits}matmul, anything else its vecgroupquant twin."""
    dev = L.require_device(x, qweight, out, scales, zeros)
    lib = L.load()
    if bits not in (2, 3, 4):
        raise L.SbqError("vecquantmatmul: only support 2/3/4 bit now")
    if x.dtype != torch.float32 or out.dtype != torch.float32 or qweight.dtype != torch.int32:
        raise L.SbqError("vecquantmatmul: x/out must be float32 and qweight int32")
    if not (x.is_contiguous() and out.is_contiguous() and qweight.is_contiguous()):
        raise L.SbqError("vecquantmatmul: tensors must be contiguous")
    if qweight.dim() != 2:
        raise L.SbqError("input2 must be with dimension == 2")
    in_f = x.shape[-1]
    batch = x.numel() // in_f
    out_f = qweight.shape[1]
    rows = (in_f + 31) // 32 * 3 if bits == 3 else (in_f * bits + 31) // 32
    if qweight.shape[0] != rows:
        raise L.SbqError("qweight has %d rows, %d-bit packing of %d input channels needs %d"
                         % (qweight.shape[0], bits, in_f, rows))
    if out.shape[-1] != out_f or out.numel() != batch * out_f:
        raise L.SbqError("output channel must be the same with input2 out_channel")
    scales = _f32c(scales, dev)
    zeros = _f32c(zeros, dev)
    groups = 1 if not group_size else (in_f + group_size - 1) // group_size
    if scales.numel() != out_f * groups or zeros.numel() != out_f * groups:
        raise L.SbqError("scales / zeros must hold out_features x groups values")
    fn = {4: lib.sbq_vecquant4matmul, 3: lib.sbq_vecquant3matmul, 2: lib.sbq_vecquant2matmul}[bits]
    with L.device_guard(dev):
        ws = _gptq_workspace(dev, lib.sbq_gptq_workspace_bytes(batch, in_f, out_f))
        rc = fn(L.ptr(x), L.ptr(qweight), L.ptr(out), L.ptr(scales), L.ptr(zeros), batch, in_f, out_f,
                int(group_size), L.ptr(ws), ws.numel(), L.stream_ptr(dev))
    L.check(rc)
    return out


def vecquantmatmul_multi(bits, x, qweights, outs, scales, zeros, group_size):
    """outs[m] += dequant(qweights[m]) @ x for up to 4 matrices that share x, in ONE launch (q / k / v; gate + up)."""
    dev = L.require_device(x, *qweights, *outs, *scales, *zeros)
    lib = L.load()
    n = len(qweights)
    if not (1 <= n <= 4 and len(outs) == n and len(scales) == n and len(zeros) == n):
        raise L.SbqError("vecquantmatmul_multi: 1 to 4 matrices with their outputs, scales and zeros")
    in_f = x.shape[-1]
    batch = x.numel() // in_f
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise L.SbqError("vecquantmatmul_multi: x must be contiguous float32")
    rows = (in_f + 31) // 32 * 3 if bits == 3 else (in_f * bits + 31) // 32
    sc = [_f32c(t, dev) for t in scales]
    zr = [_f32c(t, dev) for t in zeros]
    total = 0
    for m in range(n):
        qw, out = qweights[m], outs[m]
        if qw.dtype != torch.int32 or qw.dim() != 2 or qw.shape[0] != rows or not qw.is_contiguous():
            raise L.SbqError("vecquantmatmul_multi: qweight %d must be contiguous int32 [%d, out]" % (m, rows))
        if out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != batch * qw.shape[1]:
            raise L.SbqError("vecquantmatmul_multi: out %d must be contiguous float32 [batch, %d]" % (m, qw.shape[1]))
        total += qw.shape[1]
    arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])  # noqa: E731
    outf = (ctypes.c_int64 * n)(*[qw.shape[1] for qw in qweights])
    with L.device_guard(dev):
        need = lib.sbq_vecquantmatmul_multi_workspace_bytes(batch, in_f, n, outf)
        ws = _gptq_workspace(dev, need)
        rc = lib.sbq_vecquantmatmul_multi(int(bits), L.ptr(x), n, arr(qweights), arr(outs), arr(sc), arr(zr), outf, batch, in_f,
                                          int(group_size), L.ptr(ws), ws.numel(), L.stream_ptr(dev))
    L.check(rc)
    return outs


def gptq_mse_search(x2d, xmin, xmax, maxq, symmetric, scale, zero, norm=2.4, grid=100, n_candidates=80):
    """GPTQ find_params' grid search (quant.py:86-104) over the rows of x2d [rows, inner]; scale / zero hold the
    un-shrunk parameters and are UPDATED IN PLACE.  -> chosen candidate index per row (int32, -1: none finite)."""
    dev = L.require_device(x2d, xmin, xmax, scale, zero)
    lib = L.load()
    x2d = x2d.contiguous()
    rows, inner = x2d.shape
    for t in (xmin, xmax, scale, zero):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != rows:
            raise L.SbqError("gptq_mse_search: xmin / xmax / scale / zero must be contiguous fp32 with one value per row")
    index = torch.empty(rows, dtype=torch.int32, device=dev)
    need = lib.sbq_gptq_mse_search_workspace_bytes(rows, inner, int(n_candidates))  # long rows only: fp64 partials
    ws = torch.empty(need // 8, dtype=torch.float64, device=dev) if need else None
    with L.device_guard(dev):
        rc = lib.sbq_gptq_mse_search(L.ptr(x2d), L.dtype_id(x2d), rows, inner, L.ptr(xmin), L.ptr(xmax), int(maxq),
                                     int(bool(symmetric)), ctypes.c_float(norm), int(grid), int(n_candidates), L.ptr(scale),
                                     L.ptr(zero), L.ptr(index), L.ptr(ws), need, L.stream_ptr(dev))
    L.check(rc)
    return index


def vecquant4matmul(x, qweight, out, scales, zeros, group_size=0):
    return vecquantmatmul(4, x, qweight, out, scales, zeros, group_size)
