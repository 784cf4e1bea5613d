"""ctypes binding of libsbq.so -- the only door from Python into the HIP path.

There is deliberately NO fallback: if the shared library is missing or a symbol
declared in include/sbq.h is absent, importing/using this module raises.  The
product path never computes on the CPU (the CPU oracle lives under oracle/ and
is test infrastructure only).
"""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsbq.so")
HEADER_PATH = os.path.join(_HERE, "..", "include", "sbq.h")

# include/sbq.h enums
F32, F16, BF16 = 0, 1, 2
Q_NONE, Q_I8, Q_I32, Q_I4 = 0, 1, 2, 3
ROUND_HALF_EVEN, ROUND_HALF_UP, ROUND_HALF_DOWN = 0, 1, 2
MSE_CANDIDATES = 80
RADIX_BINS = 2048
ROWSEL_MAX = 16384
MAX_BATCH = 64
DIST_SAMPLE_WORDS = 8193  # SBQ_DIST_SAMPLE_WORDS
DIST_ROUND_WORDS = 4100  # SBQ_DIST_ROUND_WORDS

GROUP_LSQ, GROUP_Y_OFFSET = 1, 2
GROUP_BWD_CHUNK = 128

_DTYPES = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}


class GroupItem(ctypes.Structure):
    """sbq_group_item of include/sbq.h"""

    _fields_ = [
        ("x", ctypes.c_void_p),
        ("y", ctypes.c_void_p),
        ("scale", ctypes.c_void_p),
        ("zero_point", ctypes.c_void_p),
        ("mask", ctypes.c_void_p),
        ("C", ctypes.c_int64),
        ("inner", ctypes.c_int64),
        ("qmin", ctypes.c_int32),
        ("qmax", ctypes.c_int32),
        ("flags", ctypes.c_uint32),
        ("reserved", ctypes.c_uint32),
    ]

class CalibItem(ctypes.Structure):
    """sbq_calib_item of include/sbq.h"""

    _fields_ = [
        ("x", ctypes.c_void_p),
        ("C", ctypes.c_int64),
        ("inner", ctypes.c_int64),
        ("out_offset", ctypes.c_uint64),
        ("qmin", ctypes.c_int32),
        ("qmax", ctypes.c_int32),
        ("flags", ctypes.c_uint32),
        ("reserved", ctypes.c_uint32),
    ]


CALIB_SYMMETRIC = 1


class KthItem(ctypes.Structure):
    """sbq_kth_item of include/sbq.h"""

    _fields_ = [("x", ctypes.c_void_p), ("numel", ctypes.c_int64), ("k", ctypes.c_int64)]


c_i64, c_int, c_vp, c_sz, c_dbl = (
    ctypes.c_int64,
    ctypes.c_int,
    ctypes.c_void_p,
    ctypes.c_size_t,
    ctypes.c_double,
)

# name -> (restype, argtypes); mirrors include/sbq.h declaration by declaration
class GroupBwdItem(ctypes.Structure):
    """sbq_group_bwd_item of include/sbq.h"""

    _fields_ = [
        ("x", ctypes.c_void_p),
        ("scale", ctypes.c_void_p),
        ("zero_point", ctypes.c_void_p),
        ("mask", ctypes.c_void_p),
        ("gx_offset", ctypes.c_uint64),
        ("gs_offset", ctypes.c_uint64),
        ("C", ctypes.c_int64),
        ("inner", ctypes.c_int64),
        ("qmin", ctypes.c_int32),
        ("qmax", ctypes.c_int32),
        ("flags", ctypes.c_uint32),
        ("want_gs", ctypes.c_int32),
        ("gs_ratio", ctypes.c_float),
        ("reserved", ctypes.c_uint32),
    ]


_SIGNATURES = {
    "sbq_version": (c_int, []),
    "sbq_strerror": (ctypes.c_char_p, [c_int]),
    "sbq_last_hip_error": (ctypes.c_char_p, []),
    "sbq_set_tuning": (c_int, [c_int, c_int]),
    "sbq_quant_pertensor_forward": (
        c_int,
        [c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_i64, c_int, c_int, c_int, c_vp],
    ),
    "sbq_quant_perchannel_forward": (
        c_int,
        [c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_vp],
    ),
    "sbq_quant_perchannel_forward_batched": (c_int, [c_vp, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_int, c_int, c_vp]),
    "sbq_quant_lsq_forward": (c_int, [c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_int, c_vp]),
    "sbq_quant_lsq_backward": (c_int, [c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_int,
                                       ctypes.c_float, c_vp, c_sz, c_vp]),
    "sbq_dequantize_linear": (c_int, [c_vp, c_int, c_int, c_vp, c_int, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp]),
    "sbq_radix_select_workspace_bytes": (c_sz, [c_i64, c_int]),
    "sbq_percentile_select": (c_int, [c_vp, c_vp, c_int, c_int, c_i64, c_i64, c_dbl, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "sbq_kth_value": (c_int, [c_vp, c_int, c_i64, c_int, c_i64, c_vp, c_vp, c_sz, c_vp]),
    "sbq_percentile_ranks": (c_int, [c_vp, c_i64, c_int, c_dbl, c_vp, c_vp, c_vp]),
    "sbq_group_table_build": (c_int, [c_vp, c_int, c_vp, c_sz, c_vp, c_vp]),
    "sbq_quant_group_forward": (c_int, [c_vp, c_int, ctypes.c_uint32, c_int, c_int, c_int, c_vp, c_vp]),
    "sbq_group_bwd_table_build": (c_int, [c_vp, c_int, c_vp, c_sz, c_vp, c_vp, c_vp, c_vp]),
    "sbq_quant_group_backward": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "sbq_group_kth_workspace_bytes": (c_sz, [c_int]),
    "sbq_group_kth_value": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_sz, c_vp]),
    "sbq_calib_table_build": (c_int, [c_vp, c_int, c_vp, c_sz, c_vp, c_vp, c_vp]),
    "sbq_group_minmax_qparams": (c_int, [c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "sbq_group_mse_qparams": (c_int, [c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "sbq_mask_quant_forward": (
        c_int,
        [c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_vp],
    ),
    "sbq_backward_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64]),
    "sbq_quant_pertensor_backward": (
        c_int,
        [c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_int, c_vp, c_sz, c_vp],
    ),
    "sbq_quant_perchannel_backward": (
        c_int,
        [c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_vp, c_sz, c_vp],
    ),
    "sbq_stats_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64]),
    "sbq_observe_quant_perchannel_forward": (
        c_int,
        [c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_int, c_vp, c_sz, c_vp],
    ),
    "sbq_channel_stats": (c_int, [c_vp, c_int, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "sbq_channel_moments": (c_int, [c_vp, c_int, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "sbq_aciq_thresholds": (c_int, [c_vp, c_vp, c_vp, c_i64, ctypes.c_float, ctypes.c_float, ctypes.c_float, c_int, c_vp, c_vp, c_vp]),
    "sbq_minmax_pack": (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp]),
    "sbq_minmax_unpack": (c_int, [c_vp, c_i64, c_vp, c_vp, c_vp]),
    "sbq_ema_minmax": (c_int, [c_vp, c_vp, c_i64, ctypes.c_float, ctypes.c_float, c_vp, c_int, c_vp]),
    "sbq_qparams_from_minmax": (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    "sbq_lsq_init_scale": (c_int, [c_vp, c_i64, c_dbl, c_int, c_vp, c_vp]),
    "sbq_mse_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64]),
    "sbq_mse_accumulate": (
        c_int,
        [c_vp, c_int, c_i64, c_i64, c_i64, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_sz, c_vp],
    ),
    "sbq_minmax_state_reset": (c_int, [c_vp, c_vp]),
    "sbq_minmax_accumulate": (c_int, [c_vp, c_int, c_i64, c_vp, c_vp]),
    "sbq_minmax_state_read": (c_int, [c_vp, c_vp, c_vp, c_vp]),
    "sbq_mse_select": (c_int, [c_vp, c_dbl, c_vp, c_vp, c_i64, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "sbq_mse_select_devcount": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "sbq_percentile_rows": (c_int, [c_vp, c_int, c_i64, c_i64, c_dbl, c_vp, c_vp, c_vp]),
    "sbq_radix_histogram": (c_int, [c_vp, c_int, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    "sbq_radix_advance": (c_int, [c_vp, c_i64, c_int, c_int, c_vp, c_vp]),
    "sbq_radix_finish": (c_int, [c_vp, c_i64, c_int, c_int, c_vp, c_vp]),
    "sbq_sign_counts": (c_int, [c_vp, c_int, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp]),
    "sbq_mask_from_threshold": (c_int, [c_vp, c_int, c_i64, c_vp, c_vp, c_vp]),
    "sbq_gptq_mse_search": (c_int, [c_vp, c_int, c_i64, c_i64, c_vp, c_vp, c_int, c_int, ctypes.c_float, c_int, c_int, c_vp, c_vp,
                                    c_vp, c_vp, c_sz, c_vp]),
    "sbq_gptq_mse_search_workspace_bytes": (c_sz, [c_i64, c_i64, c_int]),
    "sbq_vecquantmatmul_multi": (c_int, [c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_sz, c_vp]),
    "sbq_dist_select_workspace_bytes": (c_sz, []),
    "sbq_dist_select_sample": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp]),
    "sbq_dist_select_plan": (c_int, [c_vp, c_int, c_int, c_int, c_dbl, c_i64, c_i64, c_vp, c_sz, c_vp]),
    "sbq_dist_select_sweep": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_sz, c_vp, c_vp]),
    "sbq_dist_select_advance": (c_int, [c_vp, c_int, c_int, c_int, c_dbl, c_vp, c_sz, c_vp, c_vp, c_vp, c_vp]),
    "sbq_workspace_release": (c_int, [c_vp, c_sz]),
    "sbq_gptq_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64]),
    "sbq_vecquantmatmul_multi_workspace_bytes": (c_sz, [c_i64, c_i64, c_int, c_vp]),
    "sbq_vecquant4matmul": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_sz, c_vp]),
    "sbq_vecquant3matmul": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_sz, c_vp]),
    "sbq_vecquant2matmul": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_sz, c_vp]),
}

_lib = None


def header_symbols():
    """Every function name declared in include/sbq.h (used by the CPU tests)."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sbq_[a-z0-9_]+)\s*\(", text)))


def load(strict=True):
    """dlopen libsbq.so and type every entry point.  Raises if anything is missing.

    strict=False (kernel-development tools only) tolerates entry points that are
    not built yet; the product modules always load strictly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libsbq.so not found at %s: build it with `python -m sparsebit_amd.build` "
            "(there is no CPU fallback for the HIP path)" % LIB_PATH
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            if strict:
                raise RuntimeError("libsbq.so does not export %s (declared in include/sbq.h)" % name)
            continue
        fn.restype = res
        fn.argtypes = args
    if lib.sbq_version() != 100:
        raise RuntimeError("libsbq.so version mismatch: %d" % lib.sbq_version())
    if strict:
        _lib = lib
    return lib


class SbqError(RuntimeError):
    pass


def check(status):
    if status != 0:
        lib = load()
        msg = lib.sbq_strerror(status).decode()
        hip = lib.sbq_last_hip_error().decode()
        raise SbqError("libsbq: %s (status %d%s)" % (msg, status, ", HIP " + hip if hip and status == 6 else ""))


def fresh_workspace(nbytes, device):
    """A zero-contract workspace (include/sbq.h 5b) for the selection engine / the GPTQ mat-vec: new memory, and the
    library is told to forget whatever it knew about that address range (torch's caching allocator hands freed
    addresses out again: a stale registry entry would bind the new buffer to the old one's stream)."""
    buf = torch.zeros(int(nbytes), dtype=torch.uint8, device=device)
    load().sbq_workspace_release(ctypes.c_void_p(buf.data_ptr()), buf.numel())
    return buf


def dtype_id(t):
    try:
        return _DTYPES[t.dtype]
    except KeyError:
        # same failure class as the reference's CheckTensor (common.cuh:45-49)
        raise SbqError("libsbq: Kernel Failure, Invalid dtype of Input tensor: %s" % t.dtype)


def require_device(*tensors):
    """The HIP path only: every operand must already live in HBM."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise SbqError(
                "sparsebit_amd runs on MI355X only: got a %s tensor (no CPU fallback exists; "
                "the CPU oracle is under oracle/ and is for tests)" % t.device
            )
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise SbqError("input, scale and zero_point of quantizer must be on same device!")
    return dev


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr(device=None):
    """hipStream_t of torch's current stream on `device` (the raw-handle accessor is ~10x cheaper than
    building a torch.cuda.Stream object; the quantizer calls are host-bound on small tensors)"""
    if _raw_stream is not None and device is not None and device.index is not None:
        return ctypes.c_void_p(_raw_stream(device.index))
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _NoGuard:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def device_guard(device):
    """`with device_guard(dev):` == torch.cuda.device(dev), free when dev is already current"""
    if device.index is None or torch.cuda.current_device() == device.index:
        return _NO_GUARD
    return torch.cuda.device(device)


TUNING_PROCESS = 0x100  # SBQ_TUNING_PROCESS


def set_tuning(knob, value, process=False):
    """per calling thread by default; process=True sets the default every thread without a setting of its own sees
    (autograd's backward threads never see the main thread's per-thread setting)"""
    check(load().sbq_set_tuning(knob | (TUNING_PROCESS if process else 0), value))
