"""Fused min-max observer + qparams + QDQ of a per-channel weight (csrc/sbq_qdq_resident.hip: qdq_observe_kernel)
against the three separate library steps (bit for bit, as the header promises), the oracle (the reference's CPU
arithmetic: observers/minmax.py:14-25, observers/base.py:63-79, quant_tensor.py:182-184) and through
Quantizer.calibrate_forward.  Shapes: rows of 4096 (two slabs) and 2048 (one slab) -- the fused kernel -- and the
geometries that take the three-step route inside the same entry point."""
import numpy as np
import pytest
import torch

from helpers import same_values
from sparsebit_amd import ops

pytestmark = pytest.mark.gpu


def _w(rows, inner, seed, dtype):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(rows, inner, generator=g) * torch.logspace(-2, 1, rows).unsqueeze(1)).to(dtype)


@pytest.mark.parametrize("rows,inner", [(4096, 4096), (1000, 4096), (37, 4096), (4096, 2048), (515, 2048), (64, 1024),
                                        (300, 6144), (100, 100)])
@pytest.mark.parametrize("dtype,out_dtype", [(torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32),
                                             (torch.float16, torch.float16), (torch.float32, torch.float32)])
@pytest.mark.parametrize("sym,qmin,qmax", [(True, -128, 127), (False, 0, 255), (True, -8, 7)])
def test_fused_equals_three_steps_and_oracle(rows, inner, dtype, out_dtype, sym, qmin, qmax, oracle):
    w = _w(rows, inner, rows + inner, dtype)
    if rows > 40:
        w[5] = 0.0          # zero row: scale floor 1e-6
        w[7, 3] = float("nan")  # NaN row: min = max = NaN, NaN scale -> the cold path
        w[9] = w[9].abs()   # no negatives (affine: min clamps to 0)
    wd = w.cuda()
    y, s, z, mn, mx = ops.observe_fake_quant(wd, qmin, qmax, sym, out_dtype)
    # the three steps
    mn3, mx3, _ = ops.channel_stats(wd, 0, True)
    s3, z3 = ops.qparams_from_minmax(mn3, mx3, qmin, qmax, sym)
    y3 = ops.fake_quant(wd, s3, z3, qmin, qmax, 0, out_dtype)
    assert same_values(mn.cpu().numpy(), mn3.cpu().numpy()) and same_values(mx.cpu().numpy(), mx3.cpu().numpy())
    assert same_values(s.cpu().numpy(), s3.reshape(-1).cpu().numpy())
    assert same_values(z.cpu().numpy(), z3.reshape(-1).cpu().numpy())
    assert same_values(y.float().cpu().numpy(), y3.float().cpu().numpy())
    # the oracle
    wf = w.float().numpy()
    omn, omx = oracle.minmax(wf, 0, True)
    os_, oz = oracle.qparams_from_minmax(omn, omx, qmin, qmax, sym)
    ref, _ = oracle.qdq(wf, os_, oz, qmin, qmax, 0)
    assert same_values(s.cpu().numpy(), os_) and same_values(z.cpu().numpy(), oz)
    assert same_values(y.float().cpu().numpy(), torch.from_numpy(ref).to(out_dtype).float().numpy())


def test_quantizer_calibrate_forward():
    from sparsebit_amd.common import Backend
    from sparsebit_amd.config import quantizer_config
    from sparsebit_amd.quantizers import build_quantizer

    w = _w(1024, 4096, 3, torch.float32).cuda()
    for scheme in ("per-channel-symmetric", "per-channel-affine", "per-tensor-symmetric"):
        for observer in ("MINMAX", "MSE"):
            qa = build_quantizer(quantizer_config(scheme, 8, observer=observer))
            qb = build_quantizer(quantizer_config(scheme, 8, observer=observer))
            for q in (qa, qb):
                q.set_backend(Backend.VIRTUAL)
            ya = qa.calibrate_forward(w)
            qb.update_observer(w)
            qb.calc_qparams()
            qb.enable_quant()
            yb = qb(w)
            assert torch.equal(ya, yb), (scheme, observer)
            assert torch.equal(qa.scale, qb.scale) and torch.equal(qa.zero_point, qb.zero_point)
            assert qa.scale.shape == qb.scale.shape
            if observer == "MINMAX":
                assert torch.equal(qa.observer.min_val, qb.observer.min_val)
            assert len(qa.observer.data_cache) == 0
