"""Shared helpers of the GPU parity tests."""
import numpy as np
import torch

from sparsebit_amd.common import Backend
from sparsebit_amd.config import quantizer_config


def case_config(name):
    """Rebuild the quantizer config a golden case was generated with (gen_golden.py)."""
    parts = name.split("/")
    kind, scheme = parts[0], parts[1]
    backend = Backend.TENSORRT if kind == "trt" else Backend.VIRTUAL
    tail = parts[3]
    is_act = tail.startswith(("nchw", "nlc", "relu"))
    layout = "NLC" if tail.startswith("nlc") else "NCHW"
    target = "feature" if is_act else "weight"
    if kind in ("uni", "trt", "act"):
        cfg = quantizer_config(scheme, int(parts[2]), "uniform", "MINMAX", target, layout)
    elif kind == "pct":
        cfg = quantizer_config(scheme, 8, "uniform", "PERCENTILE", target, layout, alpha=float(parts[2]))
    elif kind == "mse":
        cfg = quantizer_config(scheme, int(parts[2]), "uniform", "MSE", target, layout)
    elif kind == "lsq":
        cfg = quantizer_config(scheme, int(parts[2]), "lsq", "MINMAX", target, layout)
    else:
        raise KeyError(name)
    return cfg, backend


def all_x(golden, name):
    xs = [golden[name + "/x"]]
    i = 1
    while name + "/x%d" % i in golden:
        xs.append(golden[name + "/x%d" % i])
        i += 1
    return xs


def dev_tensor(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0").to(dtype)


def same_values(a, b):
    """bit-for-bit equal except that +0 == -0 and NaN == NaN."""
    a = np.asarray(a)
    b = np.asarray(b)
    if a.shape != b.shape:
        return False
    return bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))
