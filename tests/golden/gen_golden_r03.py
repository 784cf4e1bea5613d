"""Round-3 golden vectors from the REAL reference (authoring container only) -> ref_golden_r03.npz.

GPTQ Quantizer.find_params (large_language_models/llama/quantization/utils/quant.py:43-132) on the branches that
round 2 left unimplemented:
  * mse=True: the grid search over 80 shrink factors with the |q - x|^2.4 error (quant.py:86-104), per channel and per
    group, symmetric and asymmetric, 4 / 3 / 2 bit.  Stored besides scale / zero: the reference's error of EVERY
    candidate per row, so that a test can tell a genuine mismatch from two candidates whose errors tie to rounding
    (torch's CPU pow and the device's differ in the last bits).
  * weight=False: activations of rank 4 / 3 / 2, per channel and per tensor (quant.py:58-69, 105-132).

Re-run:  HIP_VISIBLE_DEVICES="" python tests/golden/gen_golden_r03.py
"""
import importlib
import os
import sys
import types

os.environ.setdefault("HIP_VISIBLE_DEVICES", "")
os.environ.setdefault("CUDA_VISIBLE_DEVICES", "")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

OUT = os.path.join(HERE, "ref_golden_r03.npz")


def candidate_errors(rq, qz, x2d):
    """the loop of quant.py:86-104 once more, keeping every candidate's error (same tensor ops, same order)"""
    dev = x2d.device
    tmp = torch.zeros(x2d.shape[0], device=dev)
    xmin = torch.minimum(x2d.min(1)[0], tmp)
    xmax = torch.maximum(x2d.max(1)[0], tmp)
    if qz.sym:
        xmax = torch.maximum(torch.abs(xmin), xmax)
        t = xmin < 0
        if torch.any(t):
            xmin[t] = -xmax[t]
    t = (xmin == 0) & (xmax == 0)
    xmin[t] = -1
    xmax[t] = +1
    zero0 = torch.full_like(xmax, (qz.maxq + 1) / 2)
    errs = []
    for i in range(int(qz.maxshrink * qz.grid)):
        p = 1 - i / qz.grid
        xmin1 = p * xmin
        xmax1 = p * xmax
        scale1 = (xmax1 - xmin1) / qz.maxq
        zero1 = torch.round(-xmin1 / scale1) if not qz.sym else zero0
        q = rq.quantize(x2d, scale1.unsqueeze(1), zero1.unsqueeze(1), qz.maxq)
        q -= x2d
        q.abs_()
        q.pow_(qz.norm)
        errs.append(torch.sum(q, 1))
    return torch.stack(errs, 1)


def main():
    assert not torch.cuda.is_available(), "generate goldens with GPUs hidden (reference CPU path)"
    gen_golden.install_stubs()
    llama = os.path.join(gen_golden.REF, "large_language_models/llama/quantization")
    sys.path.insert(0, llama)
    lk = types.ModuleType("utils.load_cuda_kernel")
    lk.cuda_kernel = None
    importlib.import_module("utils")
    sys.modules["utils.load_cuda_kernel"] = lk
    rq = importlib.import_module("utils.quant")

    g = torch.Generator().manual_seed(303)
    out = {}
    cases = []
    # ---- mse=True on weights ----
    weights = {"w256x384": torch.randn(24, 384, generator=g) * torch.logspace(-1, 0.5, 24).unsqueeze(1),
               "w40x136": torch.randn(40, 136, generator=g) * 0.3}
    weights["w256x384"][3] = 0.0  # an all-zero row: the (-1, +1) rule
    weights["w256x384"][5].abs_()  # a non-negative row
    for wname, w in weights.items():
        out["gptqmse/%s/w" % wname] = w.numpy()
        for bit in (4, 3, 2):
            for sym in (False, True):
                for gs in (-1, 128) if w.shape[1] % 128 == 0 else (-1,):
                    name = "gptqmse/%s/b%d/%s/g%d" % (wname, bit, "sym" if sym else "asym", gs)
                    qz = rq.Quantizer()
                    qz.configure(bit=bit, perchannel=True, sym=sym, mse=True)
                    qz.find_params(w.clone(), weight=True, groupsize=gs)
                    x2d = w.reshape(-1, gs) if gs != -1 else w.flatten(1)
                    out[name + "/scale"] = qz.scale.numpy().reshape(-1)
                    out[name + "/zero"] = qz.zero.numpy().reshape(-1)
                    out[name + "/errs"] = candidate_errors(rq, qz, x2d.clone()).numpy()
                    out[name + "/shape"] = np.array(qz.scale.shape, dtype=np.int64)
                    cases.append(name)
    # ---- weight=False ----
    acts = {"a4": torch.randn(2, 6, 5, 7, generator=g), "a3": torch.randn(3, 11, 16, generator=g), "a2": torch.randn(9, 24, generator=g)}
    acts["a4"][:, 2] = acts["a4"][:, 2].abs()
    acases = []
    for aname, a in acts.items():
        out["gptqact/%s/x" % aname] = a.numpy()
        for perch in (True, False):
            for sym in (False, True):
                for mse in (False, True):
                    name = "gptqact/%s/%s/%s/%s" % (aname, "pc" if perch else "pt", "sym" if sym else "asym", "mse" if mse else "minmax")
                    qz = rq.Quantizer()
                    qz.configure(bit=4, perchannel=perch, sym=sym, mse=mse)
                    qz.find_params(a.clone(), weight=False)
                    out[name + "/scale"] = qz.scale.numpy()
                    out[name + "/zero"] = qz.zero.numpy()
                    out[name + "/y"] = qz.quantize(a.clone()).numpy()
                    acases.append(name)
    out["cases"] = np.array(cases)
    out["act_cases"] = np.array(acases)
    np.savez_compressed(OUT, **out)
    print("wrote %s: %d arrays, %d mse cases, %d activation cases" % (OUT, len(out), len(cases), len(acases)))


if __name__ == "__main__":
    main()
