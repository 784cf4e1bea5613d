"""The resident schedule of the forward QDQ (csrc/sbq_qdq_resident.hip) against the pipelined kernels and the oracle.

The library picks the resident kernel by geometry and dtype (whole 2048-element slabs per row, outer == 1 or per
tensor, 16-bit input, about one residency of slabs), so the same call goes through BOTH kernels here: knob 3 = 1
forces the pipelined path, knob 3 = 2 the resident one (every dtype pair, whatever the size; 0 is the library's own
choice, checked to agree as well).  Results must be bit-identical to each other and to the oracle
(the reference CPU arithmetic, quant_tensor.py:182-184) -- including the last workgroup's ragged tile, the
fused masks, the LSQ pre-ops and every dtype pair.
"""
import numpy as np
import pytest
import torch

from helpers import same_values
from sparsebit_amd import lib as L
from sparsebit_amd import ops

pytestmark = pytest.mark.gpu


@pytest.fixture
def knob3():
    yield lambda v: L.set_tuning(3, v)
    L.set_tuning(3, 0)


def _weight(rows, inner, seed, dtype):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(rows, inner, generator=g) * torch.logspace(-2, 1, rows).unsqueeze(1)
    return w.to(dtype)


def _qparams(xf, qmin, qmax, sym, per_channel, oracle):
    mn, mx = oracle.minmax(xf if per_channel else xf.reshape(-1), 0, per_channel)
    return oracle.qparams_from_minmax(mn, mx, qmin, qmax, sym)


# (rows, inner): slabs = rows * inner / 2048.  4096x4096 = 8192 slabs (U = 16, full), 2050x4096 = 4100 (U = 16,
# ragged last tile), 1536x4096 = 3072 (U = 8), 1027x6144 = 3081 (U = 8, 3 slabs per row, ragged), 3000x2048 (one slab
# per row, U = 8)
# per row, U = 8), 1024x4096 = 2048 (U = 4), 601x4096 = 1202 (U = 4, ragged)
SHAPES = [(4096, 4096), (2050, 4096), (1536, 4096), (1027, 6144), (3000, 2048), (1024, 4096), (601, 4096)]


@pytest.mark.parametrize("rows,inner", SHAPES)
@pytest.mark.parametrize("dtype,out_dtype", [(torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32),
                                             (torch.float16, torch.float16), (torch.float32, torch.float32)])
@pytest.mark.parametrize("scheme", ["sym8", "affine8", "sym4"])
def test_resident_equals_pipelined_and_oracle(rows, inner, dtype, out_dtype, scheme, oracle, knob3):
    qmin, qmax, sym = {"sym8": (-128, 127, True), "affine8": (0, 255, False), "sym4": (-8, 7, True)}[scheme]
    x = _weight(rows, inner, rows + inner, dtype)
    xf = x.float().numpy()
    s, z = _qparams(xf, qmin, qmax, sym, True, oracle)
    xd, sd, zd = x.cuda(), torch.from_numpy(s).cuda(), torch.from_numpy(z).cuda()
    knob3(1)
    y_pipe = ops.fake_quant(xd, sd, zd, qmin, qmax, 0, out_dtype)
    knob3(2)
    y_res = ops.fake_quant(xd, sd, zd, qmin, qmax, 0, out_dtype)
    assert torch.equal(y_pipe.view(torch.int16 if out_dtype != torch.float32 else torch.int32),
                       y_res.view(torch.int16 if out_dtype != torch.float32 else torch.int32))
    knob3(0)
    y_auto = ops.fake_quant(xd, sd, zd, qmin, qmax, 0, out_dtype)
    assert torch.equal(y_auto.float(), y_res.float())
    ref, _ = oracle.qdq(xf, s, z, qmin, qmax, 0)
    ref = torch.from_numpy(ref).to(out_dtype).float().numpy()
    assert same_values(y_res.float().cpu().numpy(), ref)


@pytest.mark.parametrize("n", [2048 * 8192, 2048 * 5000, 2048 * 2049])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_resident_per_tensor(n, dtype, oracle, knob3):
    g = torch.Generator().manual_seed(n % 9973)
    x = (torch.randn(n, generator=g) * 3).to(dtype)
    xf = x.float().numpy()
    s, z = _qparams(xf, 0, 255, False, False, oracle)
    xd, sd, zd = x.cuda(), torch.from_numpy(s).cuda(), torch.from_numpy(z).cuda()
    knob3(1)
    y_pipe = ops.fake_quant(xd, sd, zd, 0, 255, 0, torch.float32)
    knob3(2)
    y_res = ops.fake_quant(xd, sd, zd, 0, 255, 0, torch.float32)
    assert torch.equal(y_pipe.view(torch.int32), y_res.view(torch.int32))
    ref, _ = oracle.qdq(xf, s, z, 0, 255)
    assert same_values(y_res.cpu().numpy(), ref)


@pytest.mark.parametrize("rows,inner", [(4096, 4096), (2050, 4096)])
@pytest.mark.parametrize("dtype,out_dtype", [(torch.bfloat16, torch.bfloat16), (torch.float32, torch.float32),
                                             (torch.float16, torch.float32)])
def test_resident_fused_masks_and_lsq(rows, inner, dtype, out_dtype, oracle, knob3):
    x = _weight(rows, inner, 7, dtype)
    xf = x.float().numpy()
    xd = x.cuda()
    # config 5: 50 % unstructured mask (bytes) + LSQ 4-bit with the raw (signed) learnable scale
    thresh = np.float32(np.median(np.abs(xf)))
    mask = np.abs(xf) > thresh
    s_raw = (2 * np.abs(xf).mean(axis=1) / np.sqrt(7)).astype(np.float32)
    s_raw[::3] *= -1  # LSQ pre-op: |scale|
    z_raw = np.zeros(rows, np.float32)
    z_raw[1::5] = 11.0  # LSQ pre-op: clamp(zero_point, qmin, qmax) -> 7
    sd, zd, md = torch.from_numpy(s_raw).cuda(), torch.from_numpy(z_raw).cuda(), torch.from_numpy(mask).cuda()
    knob3(1)
    y_pipe = ops.lsq_fake_quant(xd, sd, zd, -8, 7, 0, out_dtype, md)
    knob3(2)
    y_res = ops.lsq_fake_quant(xd, sd, zd, -8, 7, 0, out_dtype, md)
    assert torch.equal(y_pipe.float(), y_res.float())
    ref, _ = oracle.qdq(xf * mask, np.abs(s_raw), np.clip(z_raw, -8, 7), -8, 7, 0)
    ref = torch.from_numpy(ref).to(out_dtype).float().numpy()
    assert same_values(y_res.float().cpu().numpy(), ref)
    # threshold form of the mask (the threshold stays on the device), plain uniform quantizer
    s, z = _qparams(xf * mask, -128, 127, True, True, oracle)
    sd, zd, td = torch.from_numpy(s).cuda(), torch.from_numpy(z).cuda(), torch.tensor(thresh).cuda()
    knob3(1)
    y_pipe = ops.fake_quant(xd, sd, zd, -128, 127, 0, out_dtype, thresh=td)
    knob3(2)
    y_res = ops.fake_quant(xd, sd, zd, -128, 127, 0, out_dtype, thresh=td)
    assert torch.equal(y_pipe.float(), y_res.float())
    ref, _ = oracle.qdq(xf * mask, s, z, -128, 127, 0)
    assert same_values(y_res.float().cpu().numpy(), torch.from_numpy(ref).to(out_dtype).float().numpy())


def test_resident_edge_values(oracle, knob3):
    """ties at k + 0.5, signed zeros, denormals, values beyond the clamp, inf / NaN (the wave-vote fallback to IEEE
    division), a zero row (scale floor 1e-6) -- in a tensor the resident kernel takes."""
    rows, inner = 2048, 4096
    x = _weight(rows, inner, 3, torch.float32)
    x[5] = 0.0
    s = np.full(rows, 0.25, np.float32)
    s[5] = 1e-6
    z = np.zeros(rows, np.float32)
    x[7, :11] = torch.tensor([0.125, 0.375, 0.625, -0.125, -0.375, 31.625, 31.875, 50.0, -32.125, -32.375, -0.0])
    x[9, :4] = torch.tensor([float("inf"), float("-inf"), float("nan"), 1e-41])
    x[11] = x[11] * 1e30
    xf = x.numpy()
    xd, sd, zd = x.cuda(), torch.from_numpy(s).cuda(), torch.from_numpy(z).cuda()
    knob3(1)
    y_pipe = ops.fake_quant(xd, sd, zd, -128, 127, 0, torch.float32)
    knob3(2)
    y_res = ops.fake_quant(xd, sd, zd, -128, 127, 0, torch.float32)
    assert same_values(y_pipe.cpu().numpy(), y_res.cpu().numpy())
    ref, _ = oracle.qdq(xf, s, z, -128, 127, 0)
    assert same_values(y_res.cpu().numpy(), ref)
