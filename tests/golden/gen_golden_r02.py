"""Round-2 golden vectors from the REAL reference (authoring container only) -> ref_golden_r02.npz.

1. Per-channel MSE pinned to the reference.  The reference's own per-channel MSE is broken on the CPU
   (flat [C] scale broadcast along the wrong axis), so gen_golden.py holds per-tensor MSE
   cases only and per-channel results were pinned through the oracle's restatement.  Here every ROW of
   the `lin` (64x96) and `conv` (32x16x3x3) weights of ref_golden.npz goes through the reference's
   PER-TENSOR MSE observer (observers/mse.py:28-63) as its own tensor: per row that is exactly what
   per-channel MSE means (the CUDA kernel's row indexing, fake_quant_tensor.cu:181-186).
   Stored: scale / zero_point per row, for symmetric / affine x 8 / 4 bit.
2. Structured L1 pruning masks of the reference sparser (sparse/sparsers/l1norm.py:27-41).

Re-run:  HIP_VISIBLE_DEVICES="" python tests/golden/gen_golden_r02.py
"""
import os
import sys

os.environ.setdefault("HIP_VISIBLE_DEVICES", "")
os.environ.setdefault("CUDA_VISIBLE_DEVICES", "")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

OUT = os.path.join(HERE, "ref_golden_r02.npz")


def main():
    assert not torch.cuda.is_available(), "generate goldens with GPUs hidden (reference CPU path)"
    gen_golden.install_stubs()
    sys.path.insert(0, gen_golden.REF)
    from sparsebit.quantization.common import Backend
    from sparsebit.quantization.quantizers import build_quantizer
    from sparsebit.sparse.sparsers import build_sparser

    base = np.load(os.path.join(HERE, "ref_golden.npz"), allow_pickle=False)
    weights = {
        "lin": base["uni/per-channel-symmetric/8/lin/x"],
        "conv": base["uni/per-channel-symmetric/8/conv/x"],
    }
    out = {}
    cases = []
    for wname, w in weights.items():
        rows = torch.from_numpy(w.reshape(w.shape[0], -1).copy())
        out["rowmse/%s/x" % wname] = w
        for sym in ("symmetric", "affine"):
            for bit in (8, 4):
                name = "rowmse/per-channel-%s/%d/%s" % (sym, bit, wname)
                scales, zps = [], []
                for r in range(rows.shape[0]):
                    q = build_quantizer(gen_golden.qcfg("per-tensor-%s" % sym, bit, "MSE"))
                    q.set_backend(Backend.VIRTUAL)
                    q.update_observer(rows[r:r + 1].clone())
                    s, z = q.calc_qparams()
                    scales.append(float(s.reshape(-1)[0]))
                    zps.append(float(z.reshape(-1)[0]))
                out[name + "/scale"] = np.array(scales, dtype=np.float32)
                out[name + "/zero_point"] = np.array(zps, dtype=np.float32)
                out[name + "/meta"] = np.array([q.qdesc.qmin, q.qdesc.qmax, int(q.qdesc.is_symmetric)], dtype=np.int64)
                cases.append(name)
    # structured pruning
    C = gen_golden.CfgNode
    for wname, w in weights.items():
        for ratio in (0.25, 0.5, 0.9):
            cfg = C({"SPARSER": C({"TYPE": "structed", "STRATEGY": "l1norm", "RATIO": ratio})})
            sp = build_sparser(cfg, opr=None)
            m = sp.calc_mask(torch.from_numpy(w.copy()))
            name = "smask/%s/%g" % (wname, ratio)
            out[name + "/mask"] = m.numpy().astype(np.float32)
            cases.append(name)
    out["cases"] = np.array(cases)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, len(out), "arrays,", len(cases), "cases")


if __name__ == "__main__":
    main()
