"""numpy front-end of the C oracle (oracle/sbq_oracle.c) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module, and only as the checker / timed CPU baseline.  Every function takes
and returns numpy arrays (fp32 data; bf16/fp16 inputs are upcast by the caller,
which is the contract of SURVEY.md 9 Q1: the oracle is the reference applied to
x.float()).  Reference citations are in sbq_oracle.c next to each function.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsbq_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_f64p = ctypes.POINTER(ctypes.c_double)
_i64 = ctypes.c_int64
_int = ctypes.c_int


def build(force=False):
    src = os.path.join(_HERE, "sbq_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(src) > os.path.getmtime(_LIB_PATH):
        subprocess.run(["make", "-C", _HERE, "-B", "libsbq_oracle.so"], check=True,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        lib = ctypes.CDLL(_LIB_PATH)
        lib.orc_qdq.argtypes = [_f32p, _i64, _i64, _i64, _f32p, _f32p, _int, _int, _f32p, _i32p]
        lib.orc_mask_qdq.argtypes = [_f32p, _u8p, _i64, _i64, _i64, _f32p, _f32p, _int, _int, _f32p, _i32p]
        lib.orc_qparams_from_minmax.argtypes = [_f32p, _f32p, _i64, _int, _int, _int, _f32p, _f32p]
        lib.orc_minmax.argtypes = [_f32p, _i64, _i64, _i64, _f32p, _f32p]
        lib.orc_lsq_init_scale.argtypes = [_f32p, _i64, _i64, _i64, _int, _f32p]
        lib.orc_mse.argtypes = [_f32p, _i64, _i64, _i64, _int, _int, _int, _f32p, _f32p, _i32p, _f64p]
        lib.orc_percentile.argtypes = [_f32p, _i64, _i64, ctypes.c_double, _f32p, _f32p]
        lib.orc_l1_mask.argtypes = [_f32p, _i64, _i64, _u8p]
        lib.orc_l1_mask.restype = ctypes.c_float
        lib.orc_ste_backward.argtypes = [_f32p, _f32p, _i64, _i64, _i64, _f32p, _f32p, _int, _int, _f32p, _f32p, _f32p]
        lib.orc_vecquant4matmul.argtypes = [_f32p, _i32p, _f32p, _f32p, _f32p, _i64, _i64, _i64, _i64]
        lib.orc_vecquantmatmul.argtypes = [ctypes.c_int, _f32p, _i32p, _f32p, _f32p, _f32p, _i64, _i64, _i64, _i64]
        for f in ("orc_qdq", "orc_mask_qdq", "orc_qparams_from_minmax", "orc_minmax", "orc_lsq_init_scale",
                  "orc_mse", "orc_percentile", "orc_ste_backward", "orc_vecquant4matmul", "orc_vecquantmatmul"):
            getattr(lib, f).restype = None
        _lib = lib
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t):
    return a.ctypes.data_as(t)


def geometry(shape, ch_axis, per_channel):
    """[outer, C, inner] of a contiguous tensor (include/sbq.h)."""
    shape = tuple(int(s) for s in shape)
    if not per_channel:
        return 1, 1, int(np.prod(shape, dtype=np.int64))
    outer = int(np.prod(shape[:ch_axis], dtype=np.int64))
    inner = int(np.prod(shape[ch_axis + 1:], dtype=np.int64))
    return outer, shape[ch_axis], inner


def qdq(x, scale, zero_point, qmin, qmax, ch_axis=0, mask=None):
    """-> (dq fp32, q int32) shaped like x; scale/zero_point hold C (or 1) values."""
    x = _f32(x)
    scale = _f32(scale).reshape(-1)
    zp = _f32(zero_point).reshape(-1)
    outer, C, inner = geometry(x.shape, ch_axis, scale.size > 1)
    assert scale.size == C and zp.size == C
    dq = np.empty_like(x)
    q = np.empty(x.shape, dtype=np.int32)
    lib = _load()
    if mask is None:
        lib.orc_qdq(_p(x, _f32p), outer, C, inner, _p(scale, _f32p), _p(zp, _f32p), qmin, qmax,
                    _p(dq, _f32p), _p(q, _i32p))
    else:
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        lib.orc_mask_qdq(_p(x, _f32p), _p(m, _u8p), outer, C, inner, _p(scale, _f32p), _p(zp, _f32p),
                         qmin, qmax, _p(dq, _f32p), _p(q, _i32p))
    return dq, q


def qparams_from_minmax(min_val, max_val, qmin, qmax, symmetric):
    mn = _f32(min_val).reshape(-1)
    mx = _f32(max_val).reshape(-1)
    s = np.empty_like(mn)
    z = np.empty_like(mn)
    _load().orc_qparams_from_minmax(_p(mn, _f32p), _p(mx, _f32p), mn.size, qmin, qmax, int(symmetric),
                                    _p(s, _f32p), _p(z, _f32p))
    return s, z


def minmax(x, ch_axis=0, per_channel=True):
    x = _f32(x)
    outer, C, inner = geometry(x.shape, ch_axis, per_channel)
    mn = np.empty(C, dtype=np.float32)
    mx = np.empty(C, dtype=np.float32)
    _load().orc_minmax(_p(x, _f32p), outer, C, inner, _p(mn, _f32p), _p(mx, _f32p))
    return mn, mx


def lsq_init_scale(x, qmax, ch_axis=0, per_channel=True):
    x = _f32(x)
    outer, C, inner = geometry(x.shape, ch_axis, per_channel)
    s = np.empty(C, dtype=np.float32)
    _load().orc_lsq_init_scale(_p(x, _f32p), outer, C, inner, qmax, _p(s, _f32p))
    return s


def mse(x, qmin, qmax, symmetric, ch_axis=0, per_channel=True):
    """-> (scale, zero_point, best_index, sse[C][80])"""
    x = _f32(x)
    outer, C, inner = geometry(x.shape, ch_axis, per_channel)
    s = np.empty(C, dtype=np.float32)
    z = np.empty(C, dtype=np.float32)
    b = np.empty(C, dtype=np.int32)
    sse = np.empty((C, 80), dtype=np.float64)
    _load().orc_mse(_p(x, _f32p), outer, C, inner, qmin, qmax, int(symmetric), _p(s, _f32p), _p(z, _f32p),
                    _p(b, _i32p), _p(sse, _f64p))
    return s, z, b, sse


def mse_index_disagreements(x_rows, idx_a, idx_b, qmin, qmax, symmetric, rel=2e-6):
    """Two argmin indices per row of the MSE search (observers/mse.py:46-61) that were found with DIFFERENT summation
    trees may name different candidates where two losses tie to the rounding of an fp32 mean (the reference's own
    answer on such a row depends on torch's summation order, i.e. on its thread count).  -> the rows where idx_a and
    idx_b differ AND this oracle's fp64 sums of the two candidates differ by more than `rel` (relative): real
    disagreements.  x_rows: [R, inner] fp32 (only the rows that differ are evaluated)."""
    x_rows = _f32(x_rows)
    a = np.asarray(idx_a).reshape(-1)
    b = np.asarray(idx_b).reshape(-1)
    bad = []
    for r in np.nonzero(a != b)[0]:
        if a[r] < 0 or b[r] < 0:
            bad.append(int(r))
            continue
        _, _, _, sse = mse(x_rows[r:r + 1], qmin, qmax, symmetric, 0, True)
        la, lb = float(sse[0, a[r]]), float(sse[0, b[r]])
        if not abs(la - lb) <= rel * max(abs(la), abs(lb)):
            bad.append(int(r))
    return bad


def channel_first(x, ch_axis):
    """DataCache.get_data_for_calibration(CHANNELWISE) for ONE cached tensor
    (observers/base.py:27-31): [C, everything else]."""
    x = np.asarray(x)
    if ch_axis != 0:
        x = np.swapaxes(x, 0, ch_axis)
    return np.ascontiguousarray(x.reshape(x.shape[0], -1))


def percentile(x, alpha, ch_axis=0, per_channel=True):
    x = _f32(x)
    rows = channel_first(x, ch_axis) if per_channel else x.reshape(1, -1)
    rows = _f32(rows)
    C, n = rows.shape
    mn = np.empty(C, dtype=np.float32)
    mx = np.empty(C, dtype=np.float32)
    _load().orc_percentile(_p(rows, _f32p), C, n, float(alpha), _p(mn, _f32p), _p(mx, _f32p))
    return mn, mx


def l1_mask(x, ratio):
    """-> (mask bool shaped like x, threshold).  l1norm.py:18-26"""
    x = _f32(x)
    n = x.size
    if ratio == 0.0:
        return np.ones(x.shape, dtype=bool), None
    idx = min(int(n * ratio), n - 1)
    m = np.empty(x.shape, dtype=np.uint8)
    t = _load().orc_l1_mask(_p(x, _f32p), n, idx, _p(m, _u8p))
    return m.astype(bool), np.float32(t)


def ste_backward(x, gy, scale, zero_point, qmin, qmax, ch_axis=0):
    x = _f32(x)
    gy = _f32(gy)
    scale = _f32(scale).reshape(-1)
    zp = _f32(zero_point).reshape(-1)
    outer, C, inner = geometry(x.shape, ch_axis, scale.size > 1)
    gx = np.empty_like(x)
    gs = np.empty(C, dtype=np.float32)
    gz = np.empty(C, dtype=np.float32)
    _load().orc_ste_backward(_p(x, _f32p), _p(gy, _f32p), outer, C, inner, _p(scale, _f32p), _p(zp, _f32p),
                             qmin, qmax, _p(gx, _f32p), _p(gs, _f32p), _p(gz, _f32p))
    return gx, gs, gz


def vecquant4matmul(x, qweight, bias, scales, zeros, group_size):
    """y = bias + dequant(qweight) @ x ; x [B, in], qweight int32 [ceil(in/8), out]."""
    return vecquantmatmul(x, qweight, bias, scales, zeros, group_size, 4)


def vecquantmatmul(x, qweight, bias, scales, zeros, group_size, bits):
    """y = bias + dequant(qweight) @ x for 4 / 3 / 2-bit packed weights; x [B, in], qweight int32 [rows, out]."""
    x = _f32(x)
    B, in_f = x.shape
    qw = np.ascontiguousarray(qweight, dtype=np.int32)
    out_f = qw.shape[1]
    sc = _f32(scales).reshape(out_f, -1)
    zr = _f32(zeros).reshape(out_f, -1)
    out = np.ascontiguousarray(np.broadcast_to(_f32(bias), (B, out_f))).copy()
    _load().orc_vecquantmatmul(int(bits), _p(x, _f32p), _p(qw, _i32p), _p(out, _f32p), _p(sc, _f32p), _p(zr, _f32p),
                               B, in_f, out_f, 0 if group_size in (-1, 0) else group_size)
    return out


# ---- GPTQ host-side helpers (numpy restatements; offline steps of config 4) ----------
def gptq_find_params(w, bit=4, groupsize=-1, sym=False, mse=False, norm=2.4, grid=100, maxshrink=0.8):
    """Quantizer.find_params(weight=True, perchannel=True) (llama/quantization/utils/quant.py:43-132): per (row,
    group) min/max parameters, symmetric (:77-81, zero = (maxq + 1) / 2) or asymmetric, and optionally the `mse` grid
    search of :86-104 (80 shrink factors p = 1 - i / grid, error sum |quantize(x) - x|^norm, first strictly smaller
    wins).  -> scale, zero shaped [out, groups]  (+ errs [rows, candidates] when mse, for tie analysis)"""
    w = _f32(w)
    out_f, in_f = w.shape
    gs = in_f if groupsize == -1 else groupsize
    assert in_f % gs == 0
    maxq = np.float32(2 ** bit - 1)
    xg = w.reshape(-1, gs)
    xmin = np.minimum(xg.min(1), np.float32(0))
    xmax = np.maximum(xg.max(1), np.float32(0))
    if sym:
        xmax = np.maximum(np.abs(xmin), xmax)
        xmin = np.where(xmin < 0, -xmax, xmin).astype(np.float32)
    both0 = (xmin == 0) & (xmax == 0)
    xmin[both0] = -1
    xmax[both0] = +1
    scale = ((xmax - xmin) / maxq).astype(np.float32)
    zero = (np.full_like(scale, (maxq + 1) / 2) if sym else np.rint(-xmin / scale)).astype(np.float32)
    if not mse:
        return scale.reshape(out_f, -1), zero.reshape(out_f, -1)
    best = np.full(xg.shape[0], np.inf, dtype=np.float32)
    zero0 = zero.copy()
    errs = []
    for i in range(int(maxshrink * grid)):
        p = np.float32(1 - i / grid)
        xmin1, xmax1 = p * xmin, p * xmax
        scale1 = ((xmax1 - xmin1) / maxq).astype(np.float32)
        zero1 = zero0 if sym else np.rint(-xmin1 / scale1).astype(np.float32)
        q = np.clip(np.rint(xg / scale1[:, None]) + zero1[:, None], 0, maxq).astype(np.float32)
        d = np.abs(scale1[:, None] * (q - zero1[:, None]) - xg).astype(np.float32)
        err = np.power(d, np.float32(norm)).astype(np.float32).sum(1, dtype=np.float32)
        errs.append(err)
        better = err < best
        best = np.where(better, err, best)
        scale = np.where(better, scale1, scale)
        zero = np.where(better, zero1, zero)
    return scale.reshape(out_f, -1), zero.reshape(out_f, -1), np.stack(errs, 1)


def gptq_quantize(w, scale, zero, bit=4):
    """quantize() (quant.py:8-10) applied per group."""
    w = _f32(w)
    out_f, in_f = w.shape
    groups = scale.shape[1]
    wg = w.reshape(out_f, groups, -1)
    maxq = np.float32(2 ** bit - 1)
    s = scale[:, :, None].astype(np.float32)
    z = zero[:, :, None].astype(np.float32)
    q = np.clip(np.rint(wg / s) + z, 0, maxq).astype(np.float32)
    return (s * (q - z)).reshape(out_f, in_f).astype(np.float32)


def gptq_rows(in_f, bits):
    """qweight rows for in_f input channels (QuantLinear.__init__, quant.py:171-183)."""
    par = 3 if bits == 3 else 1
    return -(-(in_f * bits) // (32 * par)) * par


def gptq_pack(w_q, scale, zero, bits):
    """QuantLinear.pack (quant.py:187-260) for 4 / 3 / 2 bits, restated as what its loop builds:
    zeros' = zero*scale; intweight = round((w + zeros')/scale); every output column is one
    little-endian bit stream over the rows, `bits` bits per input channel (for 3 bits the
    reference's two split levels per 32 are the ones that straddle a word of that stream).
    -> qweight int32 [rows, out], zeros' [out, groups]"""
    out_f, in_f = w_q.shape
    groups = scale.shape[1]
    zeros_p = (zero * scale).astype(np.float32)
    wg = _f32(w_q).reshape(out_f, groups, -1)
    iw = np.rint((wg + zeros_p[:, :, None]) / scale[:, :, None]).astype(np.int64).reshape(out_f, in_f)
    iw = iw.T.astype(np.uint64) & np.uint64(2 ** bits - 1)  # [in, out]
    qw = np.zeros((gptq_rows(in_f, bits), out_f), dtype=np.uint64)
    for j in range(in_f):
        bit = bits * j
        idx, sh = bit >> 5, bit & 31
        qw[idx] |= (iw[j] << np.uint64(sh)) & np.uint64(0xFFFFFFFF)
        if sh + bits > 32:
            qw[idx + 1] |= iw[j] >> np.uint64(32 - sh)
    return qw.astype(np.uint32).astype(np.int32), zeros_p


def gptq_pack4(w_q, scale, zero):
    """QuantLinear.pack for bit=4 (quant.py:187-229): zeros' = zero*scale;
    intweight = round((w + zeros')/scale); 8 input-channel nibbles per int32, low first.
    -> qweight int32 [ceil(in/8), out], zeros' [out, groups]"""
    out_f, in_f = w_q.shape
    groups = scale.shape[1]
    zeros_p = (zero * scale).astype(np.float32)
    wg = _f32(w_q).reshape(out_f, groups, -1)
    iw = np.rint((wg + zeros_p[:, :, None]) / scale[:, :, None]).astype(np.int64).reshape(out_f, in_f)
    iw = iw.T.astype(np.uint32)  # [in, out]
    H = (in_f * 4 + 31) // 32
    qw = np.zeros((H, out_f), dtype=np.uint32)
    for j in range(in_f):
        qw[j // 8] |= (iw[j] & np.uint32(0xF)) << np.uint32(4 * (j % 8))
    return qw.astype(np.int32), zeros_p
