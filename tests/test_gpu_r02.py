"""Round-2 parity pins on the BASELINE.json configs (VERDICT r01 "Next round" 1).

 (a) config 2: the 32x256x56x56 per-tensor MSE activation against the oracle (index, scale, zero point);
 (b) per-channel MSE against the REFERENCE itself, row by row (tests/golden/gen_golden_r02.py);
 (c) config 1 at ResNet-18 sizes against the oracle;
 (d) the reference's own GPTQ kernel tests AS WRITTEN
     (large_language_models/llama/quantization/test_cuda_kernel.py:21-126: 13 functions x 3 bit widths).
"""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT
from helpers import same_values

pytestmark = pytest.mark.gpu


def _build(scheme, bit, observer="MINMAX", target="weight", layout="NCHW", quantizer="uniform", backend=None):
    from sparsebit_amd.common import Backend
    from sparsebit_amd.config import quantizer_config
    from sparsebit_amd.quantizers import build_quantizer

    q = build_quantizer(quantizer_config(scheme, bit, quantizer=quantizer, observer=observer, target=target, layout=layout))
    q.set_backend(backend or Backend.VIRTUAL)
    return q


# ---- (a) config 2 activation leg ----------------------------------------------------------------------
def test_config2_activation_mse_equals_oracle(oracle):
    """ResNet-50 per-tensor MSE over a 32x256x56x56 activation (25.7 M elements, four cached batches): the
    candidate the HIP path picks, its scale and zero point == oracle.mse on the same data.  This is where an
    fp32-vs-fp64 summation order could flip the argmin (the kernel sums fp32 per lane, fp64 across)."""
    g = torch.Generator().manual_seed(50)
    a = torch.relu(torch.randn(32, 256, 56, 56, generator=g))
    for scheme, (qmin, qmax, sym) in (("per-tensor-affine", (0, 255, False)), ("per-tensor-symmetric", (-128, 127, True))):
        qa = _build(scheme, 8, "MSE", "feature")
        for chunk in a.chunk(4):
            qa.update_observer(chunk.cuda())
        sa, za = qa.calc_qparams()
        rs, rz, rbest, sse = oracle.mse(a.numpy().reshape(-1), qmin, qmax, sym, per_channel=False)
        best = int(qa.observer.best_index.item())
        assert best == int(rbest[0]), (best, int(rbest[0]), sse[0, max(best - 1, 0):best + 2])
        assert np.array_equal(sa.reshape(-1).cpu().numpy(), rs) and same_values(za.reshape(-1).cpu().numpy(), rz)
        qa.enable_quant()
        ya = qa(a[:2].cuda())
        ref, _ = oracle.qdq(a[:2].numpy(), rs, rz, qmin, qmax)
        assert same_values(ya.cpu().numpy(), ref)


# ---- (b) per-channel MSE vs the reference, row by row -----------------------------------------------
def _r02():
    return np.load(os.path.join(ROOT, "tests", "golden", "ref_golden_r02.npz"), allow_pickle=False)


def _r02_cases(prefix):
    return [c for c in _r02()["cases"].tolist() if c.startswith(prefix)]


@pytest.mark.parametrize("name", _r02_cases("rowmse/"))
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_perchannel_mse_equals_reference_row_by_row(name, dtype):
    """sbq_mse_accumulate + sbq_mse_select per channel == the reference's per-tensor MSE observer
    (observers/mse.py:28-63) run on each row as its own tensor: scale bit for bit, zero point equal.
    (The weights are bf16-representable, so the bf16 run sees the same values.)"""
    z = _r02()
    _, scheme, bit, wname = name.split("/")
    x = torch.from_numpy(z["rowmse/%s/x" % wname].copy()).cuda().to(dtype)
    q = _build(scheme, int(bit), "MSE")
    q.update_observer(x)
    s, zp = q.calc_qparams()
    assert np.array_equal(s.reshape(-1).cpu().numpy(), z[name + "/scale"]), name
    assert same_values(zp.reshape(-1).cpu().numpy(), z[name + "/zero_point"]), name


@pytest.mark.parametrize("name", _r02_cases("smask/"))
def test_structured_l1_mask_equals_reference(name):
    """sparse/sparsers/l1norm.py:27-41 (TYPE 'structed'): the int(C*ratio) rows with the smallest sum|w| zeroed."""
    from sparsebit_amd.config import sparser_config
    from sparsebit_amd.sparsers import build_sparser

    z = _r02()
    _, wname, ratio = name.split("/")
    w = torch.from_numpy(z["rowmse/%s/x" % wname].copy()).cuda()
    m = build_sparser(sparser_config(float(ratio), type_="structed")).calc_mask(w)
    assert m.dtype == w.dtype and m.shape == w.shape
    assert np.array_equal(m.cpu().numpy(), z[name + "/mask"])


# ---- (c) config 1: ResNet-18 PTQ 8w8a MinMax (examples/post_training_quantization/imagenet1k/basecase) ------
def test_config1_resnet18_minmax_shapes(oracle):
    """qconfig.yaml of the basecase: TensorRT backend, W per-channel-symmetric int8, A per-tensor-symmetric
    int8, MINMAX observers, calibration batch 256.  Every tensor class of ResNet-18 at its real size: the
    256x64x56x56 activation (51.4 M elements, the layer1 input), the largest conv 512x512x3x3, the fc
    1000x512 and the 7x7 stem 64x3x7x7 (rows of 147 elements: the ragged path).  min/max, scale, zero point
    and the WHOLE dequantized tensor against the oracle."""
    from sparsebit_amd.common import Backend

    g = torch.Generator().manual_seed(18)
    a = torch.randn(256, 64, 56, 56, generator=g)
    a[0, 0, 0, :4] = torch.tensor([0.0, -0.0, 1e-30, -7.5])
    qa = _build("per-tensor-symmetric", 8, "MINMAX", "feature", backend=Backend.TENSORRT)
    for chunk in a.chunk(4):  # CalibrationRunner feeds batch by batch (tools/calibration.py:109-123)
        qa.update_observer(chunk.cuda())
    sa, za = qa.calc_qparams()
    mn, mx = oracle.minmax(a.numpy().reshape(-1), 0, False)
    assert float(qa.observer.min_val) == float(mn[0]) and float(qa.observer.max_val) == float(mx[0])
    rs, rz = oracle.qparams_from_minmax(mn, mx, -128, 127, True)
    assert np.array_equal(sa.reshape(-1).cpu().numpy(), rs) and same_values(za.reshape(-1).cpu().numpy(), rz)
    qa.enable_quant()
    ya = qa(a.cuda())
    ref, _ = oracle.qdq(a.numpy(), rs, rz, -128, 127)
    assert same_values(ya.cpu().numpy(), ref)
    del ya, ref
    for shape in ((512, 512, 3, 3), (1000, 512), (64, 3, 7, 7), (64, 64, 3, 3), (128, 64, 1, 1)):
        w = torch.randn(*shape, generator=g) * torch.logspace(-2, 0, shape[0]).reshape(-1, *([1] * (len(shape) - 1)))
        qw = _build("per-channel-symmetric", 8, backend=Backend.TENSORRT)
        qw.update_observer(w.cuda())
        s, zp = qw.calc_qparams()
        mn, mx = oracle.minmax(w.numpy(), 0, True)
        rs, rz = oracle.qparams_from_minmax(mn, mx, -128, 127, True)
        assert np.array_equal(s.reshape(-1).cpu().numpy(), rs), shape
        assert list(s.shape) == [shape[0]] + [1] * (len(shape) - 1)
        qw.enable_quant()
        y = qw(w.cuda())
        ref, _ = oracle.qdq(w.numpy(), rs, rz, -128, 127, 0)
        assert same_values(y.cpu().numpy(), ref), shape


# ---- (d) the reference's GPTQ kernel tests, as written ---------------------------------------------
# (function name, [(bit, kwargs)]) copied case for case from test_cuda_kernel.py:50-126
_KATS = {
    "OPT_175B_FC2_matvec": dict(B=1, M=12288, N=12288 * 4),
    "regular_FC": {2: dict(B=1, M=8192, N=8192 * 4), 3: dict(B=1, M=9216, N=9216 * 4), 4: dict(B=1, M=8192, N=8192 * 4)},
    "irregular_FC": dict(B=1, M=6661, N=25163),
    "single_block_regular_FC": {2: dict(B=1, M=1024, N=1024), 3: dict(B=1, M=1024, N=1024), 4: dict(B=1, M=128, N=64)},
    "single_block_irregular_FC": {2: dict(B=1, M=719, N=857), 3: dict(B=1, M=719, N=857), 4: dict(B=1, M=127, N=61)},
    "multibatch_OPT_127B_FC2_matvec": dict(B=32, M=12288, N=12288 * 4),
    "multibatch_regular_FC": dict(B=29, M=8192, N=8192 * 4),
    "multibatch_irregular_FC": dict(B=31, M=6661, N=25163),
    "multibatch_1token_FC": dict(B=32, C=1, M=6661, N=25163),
    "multibatch_8token_FC": dict(B=4, C=8, M=6661, N=25163),
    "OPT_175B_FC2_matvec_groupsize_min": {2: dict(B=1, M=12288, N=12288 * 4, GS=64), 3: dict(B=1, M=12288, N=12288 * 4, GS=128),
                                          4: dict(B=1, M=12288, N=12288 * 4, GS=128)},
    "multibatch_regular_FC_groupsize_min": {2: dict(B=29, M=8192, N=8192 * 4, GS=64), 3: dict(B=29, M=8192, N=8192 * 4, GS=128),
                                            4: dict(B=29, M=8192, N=8192 * 4, GS=128)},
    "groupsize_3x": {2: dict(B=4, M=6144, N=6144 * 4, GS=192), 3: dict(B=4, M=6144, N=6144 * 4, GS=384),
                     4: dict(B=4, M=6144, N=6144 * 4, GS=384)},
}
_KAT_CASES = [(fn, bit) for fn in _KATS for bit in (2, 3, 4)]


def run_case(bit, B, M, N, C=None, GS=-1):
    """test_cuda_kernel.py:21-45 with sparsebit_amd.gptq in place of utils.quant; the ground truth is the
    quantized nn.Linear evaluated in fp64 (the reference compares with the fp32 GEMM at the same 1e-5)."""
    from sparsebit_amd import gptq

    assert bit in [2, 3, 4]
    assert GS == -1 or GS % ({2: 64, 3: 128, 4: 128}[bit]) == 0
    torch.manual_seed(1000 * bit + B + M % 997)
    layer = torch.nn.Linear(M, N, device="cuda")
    vec = torch.randn((B, M) if C is None else (B, C, M), device="cuda")
    quantizer = gptq.Quantizer()
    quantizer.configure(bit=bit, perchannel=True, sym=False, mse=False)
    quantizer.find_params(layer.weight.data, weight=True, groupsize=GS)
    layer.weight.data = gptq.quantize(
        layer.weight.data.view(-1, M if GS == -1 else GS), quantizer.scale.view(-1, 1), quantizer.zero.view(-1, 1),
        quantizer.maxq).view(N, M)
    qlayer = gptq.QuantLinear(layer.in_features, layer.out_features, bit=bit, groupsize=GS)
    qlayer.pack(layer, quantizer.scale, quantizer.zero)
    qlayer = qlayer.to("cuda")
    with torch.no_grad():
        gt_out = torch.nn.functional.linear(vec.double(), layer.weight.double(), layer.bias.double())
        del layer
        sim_out = qlayer(vec)
        assert sim_out.shape == gt_out.shape and sim_out.dtype == torch.float32
        torch.testing.assert_close(sim_out.double(), gt_out, rtol=1e-5, atol=1e-5)
        assert torch.equal(sim_out, qlayer(vec))  # deterministic (the reference's atomicAdd order is not)
    del qlayer, gt_out, sim_out
    torch.cuda.empty_cache()


@pytest.mark.parametrize("fn,bit", _KAT_CASES, ids=["%s-%dbit" % c for c in _KAT_CASES])
def test_reference_gptq_kat(fn, bit):
    spec = _KATS[fn]
    kw = spec[bit] if bit in spec else spec
    run_case(bit=bit, **kw)


# ---- STE backward with an fp32 upstream gradient on a half-precision input ---------------------
@pytest.mark.parametrize("lsq", [False, True])
def test_backward_keeps_fp32_upstream_gradient(lsq):
    """quant_tensor.py:82-103 upcasts x and keeps the fp32 grad_y: with a bf16 activation and the default fp32
    output, the scale gradient must be the one computed from the UNROUNDED upstream gradient (== the fp32-x call),
    not from its bf16 rounding; gx comes back in bf16."""
    from sparsebit_amd import ops

    g = torch.Generator().manual_seed(4)
    x = (torch.randn(64, 96, 14, 14, generator=g) * 2).bfloat16().cuda()
    gy = torch.randn(64, 96, 14, 14, generator=g).cuda()  # fp32
    scale = torch.tensor([0.05], device="cuda")
    zp = torch.tensor([3.0], device="cuda")
    if lsq:
        gx, gs = ops.lsq_fake_quant_backward(x, gy, scale, zp, 0, 255, 0, True, 0.5)
        rx, rs = ops.lsq_fake_quant_backward(x.float(), gy, scale, zp, 0, 255, 0, True, 0.5)
        lx, ls = ops.lsq_fake_quant_backward(x, gy.bfloat16(), scale, zp, 0, 255, 0, True, 0.5)
    else:
        gx, gs, _ = ops.fake_quant_backward(x, gy, scale, zp, 0, 255, 0, True, False)
        rx, rs, _ = ops.fake_quant_backward(x.float(), gy, scale, zp, 0, 255, 0, True, False)
        lx, ls, _ = ops.fake_quant_backward(x, gy.bfloat16(), scale, zp, 0, 255, 0, True, False)
    assert gx.dtype == torch.bfloat16 and torch.equal(gx, rx.bfloat16())
    assert torch.equal(gs, rs)
    assert not torch.equal(gs, ls)  # the rounded upstream gradient gives a different sum
