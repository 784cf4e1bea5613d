"""GPU parity tests: the HIP path (through the C ABI) against
  * the golden vectors produced by the real reference (tests/golden/ref_golden.npz),
  * the CPU oracle on seeded inputs,
  * size-independent properties at BASELINE.json's full size (4096 x 4096).

Bar: integer tensor bit-exact; dequantized float bit-exact too (the contract allows 1e-6
relative; we assert 0 ulp, zeros compared by value); LSQ init / gradient sums within the
stated tolerance.
"""
import numpy as np
import pytest
import torch

from conftest import golden_cases, rel_err
from helpers import all_x, case_config, dev_tensor, same_values

pytestmark = pytest.mark.gpu

QUANT_CASES = golden_cases("")
DTYPES = [torch.float32, torch.bfloat16, torch.float16]


def _meta(golden, name):
    qmin, qmax, ch_axis, perch, sym = golden[name + "/meta"].tolist()
    return int(qmin), int(qmax), int(ch_axis), bool(perch), bool(sym)


@pytest.fixture(scope="module")
def ops():
    from sparsebit_amd import ops as o

    return o


# --------------------------------------------------------------------------------------
# forward QDQ
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", QUANT_CASES)
def test_qdq_kernel_vs_reference_golden(golden, oracle, ops, name):
    """HIP QDQ with the reference's own scale / zero_point == the reference's output."""
    qmin, qmax, ch_axis, perch, sym = _meta(golden, name)
    x = golden[name + "/x"]
    scale, zp = golden[name + "/scale"], golden[name + "/zero_point"]
    ref_dq = golden[name + "/dq"]
    _, ref_q = oracle.qdq(x, scale, zp, qmin, qmax, ch_axis)
    for dt in (torch.float32, torch.bfloat16):  # golden inputs are bf16-representable
        xd = dev_tensor(x, dt)
        y, q = ops.fake_quant(xd, dev_tensor(scale), dev_tensor(zp), qmin, qmax, ch_axis, return_q=torch.int32)
        assert y.dtype == torch.float32
        assert same_values(y.cpu().numpy(), ref_dq), (name, dt, rel_err(y.cpu().numpy(), ref_dq))
        assert np.array_equal(q.cpu().numpy(), ref_q), (name, dt)
    # perf mode: bf16 out == RNE cast of the fp32 reference result
    y16 = ops.fake_quant(dev_tensor(x, torch.bfloat16), dev_tensor(scale), dev_tensor(zp), qmin, qmax, ch_axis,
                         out_dtype=torch.bfloat16)
    want = torch.from_numpy(ref_dq).bfloat16()
    assert same_values(y16.float().cpu().numpy(), want.float().numpy())
    # int8 / uint8 storage of q
    if qmax - qmin <= 255:
        qt = torch.int8 if qmin < 0 else torch.uint8
        _, q8 = ops.fake_quant(dev_tensor(x), dev_tensor(scale), dev_tensor(zp), qmin, qmax, ch_axis, return_q=qt)
        assert np.array_equal(q8.cpu().numpy().astype(np.int32), ref_q)


@pytest.mark.parametrize("name", QUANT_CASES)
def test_quantizer_end_to_end_vs_reference_golden(golden, name):
    """build_quantizer -> update_observer -> calc_qparams -> forward, like the reference's
    calibration flow (tools/calibration.py:102-135), against the reference's own results."""
    from sparsebit_amd.quantizers import build_quantizer

    cfg, backend = case_config(name)
    qmin, qmax, ch_axis, perch, sym = _meta(golden, name)
    q = build_quantizer(cfg)
    q.set_backend(backend)
    xs = all_x(golden, name)
    for x in xs:
        q.update_observer(dev_tensor(x))
    scale, zp = q.calc_qparams()
    assert q.qdesc.qmin == qmin and q.qdesc.qmax == qmax and q.qdesc.ch_axis == ch_axis
    assert q.qdesc.is_symmetric == sym
    kind = name.split("/")[0]
    s = scale.detach().reshape(-1).cpu().numpy()
    z = zp.detach().reshape(-1).cpu().numpy()
    if kind == "lsq":
        # fp32 mean vs the reference's blocked fp32 sum: the contract's 1e-6 relative
        assert rel_err(s, golden[name + "/scale"]) <= 1e-6, rel_err(s, golden[name + "/scale"])
        assert isinstance(q.scale, torch.nn.Parameter)
        assert np.all(z == 0)
        return
    assert np.array_equal(s, golden[name + "/scale"]), (s, golden[name + "/scale"])
    assert same_values(z, golden[name + "/zero_point"])
    if kind != "mse":
        assert np.array_equal(q.observer.min_val.reshape(-1).cpu().numpy(), golden[name + "/min_val"])
        assert np.array_equal(q.observer.max_val.reshape(-1).cpu().numpy(), golden[name + "/max_val"])
    want_shape = [1] * xs[0].ndim  # Quantizer._broadcast_qparams (quantizers/base.py:97-100)
    if perch:
        want_shape[ch_axis] = golden[name + "/scale"].size
    assert list(scale.shape) == want_shape
    q.enable_quant()
    dq = q(dev_tensor(xs[0]))
    assert same_values(dq.cpu().numpy(), golden[name + "/dq"])


def test_hand_kats(golden, ops):
    x = golden["kat/x"]
    for key, s, zp, lo, hi in (("kat/int8_s1_zp0", 1.0, 0.0, -128, 127), ("kat/uint8_s1_zp3.5", 1.0, 3.5, 0, 255),
                                ("kat/uint8_s0.3_zp2.5", 0.3, 2.5, 0, 255)):
        y = ops.fake_quant(dev_tensor(x), dev_tensor([s]), dev_tensor([zp]), lo, hi)
        assert same_values(y.cpu().numpy(), golden[key]), key
    y, q = ops.fake_quant(dev_tensor(x[:11]), dev_tensor([1.0]), dev_tensor([0.0]), -128, 127, return_q=torch.int32)
    assert q.cpu().tolist() == [0, 2, 2, 0, -2, -2, 126, 127, 127, -128, -128]
    y = ops.fake_quant(dev_tensor(golden["kat/tie_x"]), dev_tensor(golden["kat/tie_s"]), dev_tensor([0.0]), -128, 127)
    assert same_values(y.cpu().numpy(), golden["kat/tie_dq"])


SHAPES = [
    ((512, 1024), 0),  # ROWS path, whole slabs
    ((96, 2304), 0),  # ROWS path, ragged last slab
    ((64, 3, 7, 7), 0),  # inner = 147: scalar path
    ((37, 40), 0),  # FLAT path (short rows)
    ((4, 24, 6, 10), 1),  # NCHW activation per channel, outer > 1
    ((3, 50, 64), 2),  # NLC per channel: channels-last pack kernel
    ((5, 33, 1536), 2),  # the same, DeiT-like width, ragged pack count per workgroup
    ((2, 7, 12), 2),  # NLC with C % 8 != 0: scalar path
    ((1, 1), 0),
]


@pytest.mark.parametrize("shape,ch_axis", SHAPES)
@pytest.mark.parametrize("scheme", ["sym8", "aff8", "sym4", "aff4"])
@pytest.mark.parametrize("dtype", DTYPES)
def test_qdq_random_vs_oracle(oracle, ops, shape, ch_axis, scheme, dtype):
    g = torch.Generator().manual_seed(sum(shape) * 131 + len(scheme) + ord(scheme[0]) + int(scheme[3:]))
    x = torch.randn(*shape, generator=g) * 3
    x = x.to(dtype)
    xf = x.float().numpy()
    qmin, qmax = {"sym8": (-128, 127), "aff8": (0, 255), "sym4": (-8, 7), "aff4": (0, 15)}[scheme]
    for perch in (True, False):
        mn, mx = oracle.minmax(xf, ch_axis, perch)
        s, z = oracle.qparams_from_minmax(mn, mx, qmin, qmax, scheme.startswith("sym"))
        if scheme.startswith("aff"):
            z = z + np.float32(0.5) * (np.arange(z.size) % 2).astype(np.float32)  # zp on .5: half-to-even rounding
        ref_dq, ref_q = oracle.qdq(xf, s, z, qmin, qmax, ch_axis)
        y, q = ops.fake_quant(x.cuda(), dev_tensor(s), dev_tensor(z), qmin, qmax, ch_axis, return_q=torch.int32)
        assert same_values(y.cpu().numpy(), ref_dq)
        assert np.array_equal(q.cpu().numpy(), ref_q)
        if dtype != torch.float32:
            y16 = ops.fake_quant(x.cuda(), dev_tensor(s), dev_tensor(z), qmin, qmax, ch_axis, out_dtype=dtype)
            assert same_values(y16.float().cpu().numpy(), torch.from_numpy(ref_dq).to(dtype).float().numpy())


def test_qdq_edge_values_and_alignment(oracle, ops):
    """ties at k+0.5, signed zeros, denormals, beyond-clamp values, inf / NaN, a zero row
    (scale floor 1e-6), odd pointer offsets, ragged per-tensor tails."""
    s = np.array([0.0123, 1e-6, 1.0, 3.7e5], dtype=np.float32)
    k = np.arange(-140, 140, dtype=np.float32)
    rows = []
    for sc in s:
        special = np.array([0.0, -0.0, 1e-40, -1e-40, 1e-30, 3e38, -3e38, np.inf, -np.inf, np.nan], np.float32)
        rows.append(np.concatenate([(k + 0.5) * sc, k * sc, special, np.zeros(6, np.float32)]))
    x = np.stack(rows).astype(np.float32)  # [4, 576]
    z = np.array([0, 0, 3.5, -2.5], dtype=np.float32)
    for (qmin, qmax) in ((-128, 127), (0, 255)):
        ref_dq, ref_q = oracle.qdq(x, s, z, qmin, qmax, 0)
        y, q = ops.fake_quant(dev_tensor(x), dev_tensor(s), dev_tensor(z), qmin, qmax, 0, return_q=torch.int32)
        assert same_values(y.cpu().numpy(), ref_dq)
        finite = ~np.isnan(x)
        assert np.array_equal(q.cpu().numpy()[finite], ref_q[finite])
    # per-tensor with a ragged tail and a misaligned base pointer (storage offset 1 element)
    g = torch.Generator().manual_seed(5)
    for n in (1, 7, 8, 9, 2049, 40003):
        for dt in DTYPES:
            base = (torch.randn(n + 3, generator=g) * 4).to(dt).cuda()
            for off in (0, 1, 3):
                xv = base[off:off + n]
                xf = xv.float().cpu().numpy()
                ref_dq, ref_q = oracle.qdq(xf, [0.031], [1.0], -128, 127)
                y, q = ops.fake_quant(xv, dev_tensor([0.031]), dev_tensor([1.0]), -128, 127, return_q=torch.int32)
                assert same_values(y.cpu().numpy(), ref_dq), (n, dt, off)
                assert np.array_equal(q.cpu().numpy(), ref_q)


def test_fast_division_equals_ieee(oracle, ops):
    """The row kernels replace the IEEE division sequence by reciprocal + two fma refinements
    (sbq_common.hpp: fast_div).  Compare with the true-division path (taken for a misaligned
    pointer) bit for bit on data built to sit on rounding boundaries, for awkward and extreme
    scales, and with the oracle on a sample."""
    g = torch.Generator().manual_seed(77)
    C, inner = 512, 4096
    # scales: random mantissas over many binades, mantissa all-ones, powers of two, range edges
    s = torch.exp2(torch.randint(-40, 30, (C,), generator=g).float()) * (1 + torch.rand(C, generator=g))
    s[0], s[1], s[2], s[3] = 2.0 ** -60, 2.0 ** 60, 2.0 ** -61, 2.0 ** 61  # in and just out of the fast range
    s[4] = torch.tensor(np.float32(np.uint32(0x3FFFFFFF).view(np.float32)).item())  # 1.9999999 (mantissa all ones)
    s[5], s[6], s[7] = 1.0, 1e-6, 3.0
    k = torch.randint(-300, 300, (C, inner), generator=g).float() + 0.5  # exact half-integers
    x = k * s[:, None]
    # nudge a third of the values by +-1 ulp around the tie, a third fully random
    ulp = torch.nextafter(x, torch.full_like(x, float("inf"))) - x
    sel = torch.randint(0, 6, x.shape, generator=g)
    x = torch.where(sel == 0, x + ulp, torch.where(sel == 1, x - ulp, x))
    x = torch.where(sel >= 4, torch.randn(x.shape, generator=g) * s[:, None] * 100, x)
    x[:, 0] = float("inf")
    x[:, 1] = float("-inf")
    x[:, 2] = float("nan")
    x[:, 3] = 3.0e38
    x[:, 4] = 1e-45
    x[:, 5] = -0.0
    z = torch.zeros(C)
    z[::3] = 7.0
    for qmin, qmax in ((-128, 127), (0, 255), (-32768, 32767)):
        for dt in (torch.float32, torch.bfloat16):
            xd = x.to(dt).cuda()
            sd, zd = s.cuda(), z.cuda()
            y_fast, q_fast = ops.fake_quant(xd, sd, zd, qmin, qmax, 0, return_q=torch.int32)
            pad = torch.empty(xd.numel() + 1, dtype=dt, device="cuda")
            xm = pad[1:].view(C, inner)  # 2- or 4-byte aligned only -> scalar kernel, true division
            xm.copy_(xd)
            y_ieee, q_ieee = ops.fake_quant(xm, sd, zd, qmin, qmax, 0, return_q=torch.int32)
            assert same_values(y_fast.cpu().numpy(), y_ieee.cpu().numpy()), (qmin, dt)
            fin = ~torch.isnan(xd.float())
            assert torch.equal(q_fast[fin], q_ieee[fin])
            rows = [0, 1, 2, 3, 4, 5, 6, 7, 100, 511]
            ref_dq, ref_q = oracle.qdq(xd.float().cpu().numpy()[rows], s.numpy()[rows], z.numpy()[rows], qmin, qmax, 0)
            assert same_values(y_fast[rows].cpu().numpy(), ref_dq)


@pytest.mark.parametrize("dtype,out_dtype", [(torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32),
                                             (torch.float32, torch.float32), (torch.float16, torch.float16)])
@pytest.mark.parametrize("shape,ch_axis,n", [((256, 1024), 0, 5), ((96, 2304), 0, 3), ((4, 24, 6, 8), 1, 2), ((8, 8), 0, 64)])
def test_batched_launch_equals_per_tensor(oracle, ops, dtype, out_dtype, shape, ch_axis, n):
    """sbq_quant_perchannel_forward_batched == n separate launches == oracle, bit for bit."""
    g = torch.Generator().manual_seed(n)
    xs = [(torch.randn(*shape, generator=g) * (i + 1)).to(dtype).cuda() for i in range(n)]
    C = shape[ch_axis]
    scales = [(torch.rand(C, generator=g) * 0.05 + 1e-3).cuda() for _ in range(n)]
    zps = [torch.randint(0, 9, (C,), generator=g).float().cuda() for _ in range(n)]
    bq = ops.BatchedFakeQuant(xs, scales, zps, 0, 255, ch_axis, out_dtype)
    outs = bq()
    for x, s, z, y in zip(xs, scales, zps, outs):
        one = ops.fake_quant(x, s, z, 0, 255, ch_axis, out_dtype=out_dtype)
        assert torch.equal(one, y)
    ref, _ = oracle.qdq(xs[-1].float().cpu().numpy(), scales[-1].cpu().numpy(), zps[-1].cpu().numpy(), 0, 255, ch_axis)
    assert same_values(outs[-1].float().cpu().numpy(), torch.from_numpy(ref).to(out_dtype).float().numpy())


def test_launches_follow_the_current_torch_stream(ops):
    """kernels go to torch's CURRENT stream (raw handle from torch._C): work queued on a side stream right
    before and after the call is ordered with it, no device-wide synchronisation needed"""
    side = torch.cuda.Stream()
    scale = torch.full((512,), 0.02, device="cuda")
    zp = torch.zeros(512, device="cuda")
    big = torch.randn(512, 8192, device="cuda")
    want = ops.fake_quant(big * 3.0 + 1.0, scale, zp, -128, 127, 0) * 2.0
    torch.cuda.synchronize()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(20):  # keep the side stream busy so that a launch on the wrong stream would race
            x = big * 3.0 + 1.0
            y = ops.fake_quant(x, scale, zp, -128, 127, 0) * 2.0
        assert L_stream_is(side)
    side.synchronize()
    assert torch.equal(y, want)


def test_kernels_are_hip_graph_capturable(ops):
    """no host synchronisation, no allocation, no default-stream work inside the library: forward, backward,
    statistics, the one-call radix select and a model-wide launch record into a HIP graph and replay"""
    g = torch.Generator().manual_seed(2)
    x = torch.randn(256, 2048, generator=g).cuda().bfloat16()
    gy = torch.randn(256, 2048, generator=g).cuda().bfloat16()
    scale = torch.full((256,), 0.03, device="cuda")
    zp = torch.zeros(256, device="cuda")
    ws = [torch.randn(s, generator=g).cuda() for s in ((64, 64, 3, 3), (128, 256), (32, 8))]
    group = ops.GroupFakeQuant([(w, torch.full((w.shape[0],), 0.05, device="cuda"), torch.zeros(w.shape[0], device="cuda"), -8, 7)
                                for w in ws])

    def work():
        y = ops.fake_quant(x, scale, zp, -128, 127, 0, out_dtype=torch.bfloat16)
        gx, gs, _ = ops.fake_quant_backward(x, gy, scale, zp, -128, 127, 0, True, False)
        mn, mx, _ = ops.channel_stats(x, 0, True)
        kth = ops.kth_value(x, 1000, use_abs=True)
        outs = [o.clone() for o in group()]
        return [y, gx, gs, mn, mx, kth] + outs

    work()  # warm up (workspaces, lazy init) outside the capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured = work()
    for rep in range(2):
        x.mul_(1.5)  # new inputs in the same storage
        for w in ws:
            w.add_(0.01)
        graph.replay()
        torch.cuda.synchronize()
        for a, b in zip(captured, work()):
            assert torch.equal(a, b)


def L_stream_is(stream):
    from sparsebit_amd import lib as L

    return L.stream_ptr(torch.device("cuda", torch.cuda.current_device())).value == (stream.cuda_stream or None)


def test_rounding_modes(ops):
    """common.cuh:64-77: half-even (0), half-up floor(v+.5) (1), half-down ceil(v-.5) (2)."""
    x = torch.tensor([0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 0.4, -0.6], device="cuda")
    one, zero = dev_tensor([1.0]), dev_tensor([0.0])
    from sparsebit_amd import fake_quant as fq

    assert fq.quant_pertensor_forward(x, one, zero, -128, 127, 0).tolist() == [0, 2, 2, 0, -2, -2, 0, -1]
    assert fq.quant_pertensor_forward(x, one, zero, -128, 127, 1).tolist() == [1, 2, 3, 0, -1, -2, 0, -1]
    assert fq.quant_pertensor_forward(x, one, zero, -128, 127, 2).tolist() == [0, 1, 2, -1, -2, -3, 0, -1]


def test_errors_and_no_cpu_fallback(ops):
    from sparsebit_amd import lib as L

    one, zero = dev_tensor([1.0]), dev_tensor([0.0])
    with pytest.raises(L.SbqError):  # CPU tensors are rejected: there is no fallback path
        ops.fake_quant(torch.randn(8), torch.ones(1), torch.zeros(1), -128, 127)
    with pytest.raises(L.SbqError):  # reference: InvalidValueException "Tensor is empty" (common.cuh:50-54)
        ops.fake_quant(torch.empty(0, device="cuda"), one, zero, -128, 127)
    with pytest.raises(L.SbqError):  # reference: ValueTypeException (common.cuh:45-49)
        ops.fake_quant(torch.zeros(8, dtype=torch.float64, device="cuda"), one, zero, -128, 127)
    with pytest.raises(L.SbqError):
        ops.fake_quant(torch.zeros(4, 8, device="cuda"), dev_tensor([1.0, 1.0]), dev_tensor([0.0, 0.0]), -128, 127, 0)
    lib = L.load()
    assert lib.sbq_quant_pertensor_forward(None, 0, None, 0, None, 0, None, None, 8, -128, 127, 0, None) == 3
    assert lib.sbq_strerror(2).decode().startswith("Kernel Failure, Tensor is empty")


# --------------------------------------------------------------------------------------
# full size (BASELINE.json: per-channel 4096 x 4096 bf16): properties + oracle on a sample
# --------------------------------------------------------------------------------------
def test_full_size_4096_properties(oracle, ops):
    g = torch.Generator().manual_seed(0)
    w = torch.randn(4096, 4096, generator=g) * torch.logspace(-2, 1, 4096).unsqueeze(1)
    x = w.bfloat16().cuda()
    from sparsebit_amd.config import quantizer_config
    from sparsebit_amd.common import Backend
    from sparsebit_amd.quantizers import build_quantizer

    q = build_quantizer(quantizer_config("per-channel-symmetric", 8))
    q.set_backend(Backend.VIRTUAL)
    q.update_observer(x)
    scale, zp = q.calc_qparams()
    # observer == oracle on all 4096 channels (cheap: one pass on the CPU)
    xf = x.float().cpu().numpy()
    mn, mx = oracle.minmax(xf, 0, True)
    s_ref, z_ref = oracle.qparams_from_minmax(mn, mx, -128, 127, True)
    assert np.array_equal(scale.reshape(-1).cpu().numpy(), s_ref)
    y, qi = ops.fake_quant(x, scale, zp, -128, 127, 0, return_q=torch.int8)
    y16 = ops.fake_quant(x, scale, zp, -128, 127, 0, out_dtype=torch.bfloat16)
    # (1) integer range, (2) dq == q * scale exactly, (3) bf16 output is the RNE cast,
    # (4) idempotence: QDQ of an fp32 QDQ result is itself, (5) every row reaches +-127 or -128
    assert int(qi.min()) >= -128 and int(qi.max()) <= 127
    assert torch.equal(y, qi.float() * scale.reshape(-1, 1))
    assert torch.equal(y16, y.bfloat16())
    y2 = ops.fake_quant(y, scale, zp, -128, 127, 0)
    assert torch.equal(y2, y)
    assert bool((qi.int().abs().amax(1) >= 127).all())  # symmetric scale: the extreme lands on 127.5 -> clamps
    # (6) the oracle on a sample of rows, bit for bit
    rows = np.array([0, 1, 17, 255, 1024, 2047, 2048, 3333, 4095])
    ref_dq, ref_q = oracle.qdq(xf[rows], s_ref[rows], z_ref[rows], -128, 127, 0)
    assert same_values(y[rows].cpu().numpy(), ref_dq)
    assert np.array_equal(qi[rows].cpu().numpy().astype(np.int32), ref_q)
    # (7) checksum of checksums against the oracle over the whole tensor
    ref_all, _ = oracle.qdq(xf, s_ref, z_ref, -128, 127, 0)
    assert same_values(y.cpu().numpy(), ref_all)


@pytest.mark.parametrize("inner", [8, 1016, 1024, 1032, 2040, 2048, 2056, 3072, 5000 * 8])
@pytest.mark.parametrize("dtype", DTYPES)
def test_split_lane_mapping_edges(oracle, ops, inner, dtype):
    """fp32 tensors give a lane two 4-element runs 1024 elements apart: row lengths around the half-block and
    block boundaries, through forward (+ mask, + levels), backward, statistics and DequantizeLinear"""
    g = torch.Generator().manual_seed(inner)
    C = 5
    x = (torch.randn(C, inner, generator=g) * 2).to(dtype).cuda()
    xf = x.float().cpu().numpy()
    scale = (torch.rand(C, generator=g) * 0.1 + 0.02).numpy()
    zp = torch.randint(0, 200, (C,), generator=g).float().numpy()
    ref_dq, ref_q = oracle.qdq(xf, scale, zp, 0, 255, 0)
    y, q = ops.fake_quant(x, dev_tensor(scale), dev_tensor(zp), 0, 255, 0, return_q=torch.uint8)  # fp32 out
    assert same_values(y.cpu().numpy(), ref_dq) and np.array_equal(q.cpu().numpy().astype(np.int32), ref_q)
    yq, q32 = ops.fake_quant(x, dev_tensor(scale), dev_tensor(zp), 0, 255, 0, return_q=torch.int32)
    assert torch.equal(yq, y) and np.array_equal(q32.cpu().numpy(), ref_q)
    m = (torch.rand(C, inner, generator=g) > 0.5).cuda()
    ref_m, _ = oracle.qdq(xf, scale, zp, 0, 255, 0, mask=m.cpu().numpy())
    assert same_values(ops.fake_quant(x, dev_tensor(scale), dev_tensor(zp), 0, 255, 0, mask=m).cpu().numpy(), ref_m)
    assert torch.equal(ops.dequantize_linear(q, dev_tensor(scale), dev_tensor(zp)), y)
    mn, mx, ab = ops.channel_stats(x, 0, True, want_abssum=True)
    omn, omx = oracle.minmax(xf, 0, True)
    assert same_values(mn.cpu().numpy(), omn) and same_values(mx.cpu().numpy(), omx)
    assert np.allclose(ab.cpu().numpy(), np.abs(xf.astype(np.float64)).sum(1), rtol=1e-6)
    gy = torch.randn(C, inner, generator=g).to(dtype).cuda()
    s8 = np.full(C, 0.05, np.float32)
    gx, gs, gz = ops.fake_quant_backward(x, gy, dev_tensor(s8), dev_tensor(zp * 0), -8, 7, 0, gx_dtype=torch.float32)
    ogx, ogs, ogz = oracle.ste_backward(xf, gy.float().cpu().numpy(), s8, zp * 0, -8, 7, 0)
    assert same_values(gx.cpu().numpy(), ogx)
    assert np.allclose(gs.cpu().numpy(), ogs, rtol=1e-5, atol=2e-5 * max(1.0, float(np.abs(ogs).max())))
    # per tensor: one long row
    ref_t, _ = oracle.qdq(xf, scale[:1], zp[:1], 0, 255, 0)
    assert same_values(ops.fake_quant(x, dev_tensor(scale[:1]), dev_tensor(zp[:1]), 0, 255, 0).cpu().numpy(), ref_t)


GROUP_SHAPES = [(64, 64, 3, 3), (8, 8), (512, 4608), (1000, 512), (3, 8), (256, 64, 1, 1), (96, 2304), (1, 16), (33, 1096)]


@pytest.mark.parametrize("dtype,out_dtype", [(torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16),
                                              (torch.bfloat16, torch.float32), (torch.float16, torch.float16)])
@pytest.mark.parametrize("masked", [False, True])
def test_group_launch_equals_per_tensor(oracle, ops, dtype, out_dtype, masked):
    """ONE launch over tensors of mixed shapes, integer ranges and granularity == fake_quant one by one
    (which the other tests pin to the oracle); plus the oracle directly on two of the items."""
    g = torch.Generator().manual_seed(17)
    entries, masks, want = [], [], []
    for i, shape in enumerate(GROUP_SHAPES):
        x = (torch.randn(shape, generator=g) * (0.5 + i)).to(dtype).cuda()
        per_channel = i % 3 != 2
        qmin, qmax = [(-8, 7), (-128, 127), (0, 255), (0, 15)][i % 4]
        C = shape[0] if per_channel else 1
        scale = (torch.rand(C, generator=g) * 0.1 + 0.01).cuda() * (0.5 + i)
        zp = torch.zeros(C).cuda() if qmin < 0 else torch.randint(0, qmax, (C,), generator=g).float().cuda()
        m = (torch.rand(shape, generator=g) > 0.4).cuda() if masked else None
        entries.append((x, scale, zp, qmin, qmax))
        masks.append(m)
        want.append(ops.fake_quant(x, scale, zp, qmin, qmax, 0, out_dtype=out_dtype, mask=m))
    gq = ops.GroupFakeQuant(entries, out_dtype=out_dtype, masks=masks if masked else None)
    assert gq.n_tiles == sum(-(-(x.numel() // 8) // 256) for x, *_ in entries)
    for rep in range(2):  # the table is reusable
        outs = gq()
        for i, (y, w) in enumerate(zip(outs, want)):
            assert y.dtype == out_dtype and torch.equal(y, w), (i, GROUP_SHAPES[i])
    for i in (2, 3):
        x, scale, zp, qmin, qmax = entries[i]
        xf = x.float()
        if masked:
            xf = xf * masks[i]
        ref, _ = oracle.qdq(xf.cpu().numpy(), scale.cpu().numpy(), zp.cpu().numpy(), qmin, qmax, 0)
        assert same_values(outs[i].float().cpu().numpy(), torch.from_numpy(ref).to(out_dtype).float().numpy())


def test_group_launch_lsq_preops_and_inplace_updates(ops):
    """SBQ_GROUP_LSQ = |scale| and clamp(zero_point) inside the kernel (lsq.py:61-62); the table holds
    pointers, so in-place updates of weights / scales are seen by the next call"""
    g = torch.Generator().manual_seed(3)
    ws = [torch.randn(s, generator=g).cuda() for s in ((32, 16, 3, 3), (64, 256), (10, 64))]
    scales = [(torch.randn(w.shape[0], generator=g) * 0.05).cuda() for w in ws]  # LSQ steps may go negative
    zps = [torch.full((w.shape[0],), 20.0).cuda() for w in ws]  # beyond qmax: clamped to 7
    gq = ops.GroupFakeQuant([(w, s, z, -8, 7) for w, s, z in zip(ws, scales, zps)], lsq=True)
    for step in range(2):
        outs = gq()
        for w, s, z, y in zip(ws, scales, zps, outs):
            assert torch.equal(y, ops.fake_quant(w, s.abs(), z.clamp(-8, 7), -8, 7, 0))
        for w, s in zip(ws, scales):  # "optimizer step", in place
            w.mul_(0.9)
            s.add_(0.01)


@pytest.mark.parametrize("masked", [False, True])
def test_weight_quant_group_forward_and_gradients(masked):
    """WeightQuantGroup == the quantizers called one by one: values, weight gradients and LSQ step-size
    gradients bit for bit, incl. tensors that fall back (7x7 conv, a disabled quantizer, LSQ+)."""
    from sparsebit_amd.common import Backend
    from sparsebit_amd.config import quantizer_config
    from sparsebit_amd.group import WeightQuantGroup
    from sparsebit_amd.quantizers import build_quantizer

    torch.manual_seed(0)
    specs = [((64, 3, 7, 7), "lsq", 8), ((64, 64, 3, 3), "lsq", 4), ((256, 64, 1, 1), "lsq", 4), ((128, 64, 3, 3), "uniform", 8),
             ((10, 256), "lsq", 8), ((32, 64, 1, 1), "lsq+", 4), ((48, 32, 3, 3), "lsq", 4)]
    triples = []
    for k, (shape, kind, bit) in enumerate(specs):
        w = torch.nn.Parameter(torch.randn(shape, device="cuda") * 0.1)
        q = build_quantizer(quantizer_config("per-channel-symmetric", bit, quantizer=kind))
        q.set_backend(Backend.VIRTUAL)
        q.update_observer(w.detach())
        q.calc_qparams()
        if k != 6:
            q.enable_quant()  # the last one stays disabled: identity
        if kind == "lsq" and k == 2:
            with torch.no_grad():
                q.scale[::2] *= -1  # learned step sizes may change sign: |scale| is what quantizes
        m = (torch.rand(shape, device="cuda") > 0.5) if masked else None
        triples.append((q, w, m))
    group = WeightQuantGroup(triples)
    assert len(group.members) == 4 and sorted(group.rest_idx) == [0, 5, 6]
    gys = [torch.randn(s, device="cuda") for s, _, _ in specs]

    def run(fn):
        for q, w, _ in triples:
            w.grad = None
            if isinstance(q.scale, torch.nn.Parameter):
                q.scale.grad = None
        outs = fn()
        loss = sum((y * gy).sum() for y, gy in zip(outs, gys))
        loss.backward()
        return ([y.detach().clone() for y in outs], [w.grad.clone() for _, w, _ in triples],
                [q.scale.grad.clone() if isinstance(q.scale, torch.nn.Parameter) and q.scale.grad is not None else None
                 for q, _, _ in triples])

    ref = run(lambda: [q(w if m is None else w * m) for q, w, m in triples])
    got = run(group)
    for k in range(len(specs)):
        assert torch.equal(got[0][k], ref[0][k]), ("value", k)
        assert torch.equal(got[1][k], ref[1][k]), ("weight grad", k)
        assert (got[2][k] is None) == (ref[2][k] is None)
        if ref[2][k] is not None:
            # the grouped backward folds 512-element segments, the per-tensor kernel 8192-element
            # chunks: same fp64 accumulation, different fp32 partial boundaries
            tol = 2e-5 * max(1.0, float(ref[2][k].abs().max()))
            assert torch.allclose(got[2][k], ref[2][k], rtol=1e-5, atol=tol), ("scale grad", k)
    with torch.no_grad():
        for y, r in zip(group(), ref[0]):
            assert torch.equal(y, r)
        # re-calibration replaces the scale tensors: the group notices and rebuilds its table
        q1, w1, m1 = triples[3]
        q1.update_observer(w1.detach() * 2)
        q1.calc_qparams()
        want = q1(w1 if m1 is None else w1 * m1)
        assert torch.equal(group()[3], want)
        # set_bit() changes the integer range baked into the device table: the group rebuilds it
        q1.set_bit(4)
        q1.update_observer(w1.detach())
        q1.calc_qparams()
        want = q1(w1 if m1 is None else w1 * m1)
        assert torch.equal(group()[3], want)
        # enabling / disabling a quantizer changes who takes part in the grouped launch
        q6, w6, m6 = triples[6]
        q6.enable_quant()
        want = q6(w6 if m6 is None else w6 * m6)
        assert torch.equal(group()[6], want) and 6 not in group.rest_idx
        q1.disable_quant()
        assert torch.equal(group()[3], w1 if m1 is None else w1 * m1) and 3 in group.rest_idx
        q1.enable_quant()
    # an in-place update of a weight between the grouped forward and its backward must not pass silently (the
    # backward kernels read the live tensors through the table)
    outs = group()
    with torch.no_grad():
        triples[1][1].add_(1.0)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        sum((y * gy).sum() for y, gy in zip(outs, gys)).backward()


@pytest.mark.parametrize("scheme,shape", [("per-channel-symmetric", (48, 32, 3, 3)), ("per-tensor-affine", (8, 16, 14, 14)),
                                          ("per-channel-symmetric", (10, 33)), ("per-tensor-symmetric", (5, 1000))])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_lsq_fused_node_equals_generic_route(scheme, shape, dtype):
    """LsqSTE (|s|, clamp(zp), gradient scaling and sign(s) inside the kernels) == abs / clamp / gs_scaling tensor
    ops around the STE: values and input gradients bit for bit, step-size gradients to summation order"""
    from sparsebit_amd.common import Backend
    from sparsebit_amd.config import quantizer_config
    from sparsebit_amd.quantizers import build_quantizer
    from sparsebit_amd.quantizers.base import Quantizer as Base

    torch.manual_seed(7)
    target = "weight" if "channel" in scheme else "feature"
    q = build_quantizer(quantizer_config(scheme, 4, quantizer="lsq", target=target))
    q.set_backend(Backend.VIRTUAL)
    x0 = torch.randn(shape, device="cuda").to(dtype)
    if "affine" in scheme:
        x0 = x0.abs()  # LSQ keeps an affine scheme only for non-negative data (lsq.py:39-43)
    q.update_observer(x0)
    q.calc_qparams()
    q.enable_quant()
    with torch.no_grad():
        q.scale.mul_(torch.where(torch.rand_like(q.scale) > 0.5, 1.0, -1.0))  # learned steps may change sign
    gy = torch.randn(shape, device="cuda").to(dtype)

    def run(fn):
        x = x0.clone().requires_grad_(True)
        q.scale.grad = None
        y = fn(x)
        y.backward(gy.to(y.dtype))
        return y.detach(), x.grad.clone(), q.scale.grad.clone()

    fused = run(lambda x: q(x))
    generic = run(lambda x: Base.forward(q, x))
    assert torch.equal(fused[0], generic[0])
    assert torch.equal(fused[1], generic[1])
    tol = 2e-5 * max(1.0, float(generic[2].abs().max()))
    assert torch.allclose(fused[2], generic[2], rtol=1e-5, atol=tol)


def test_weight_quant_group_attach_runs_inside_unmodified_operators(ops, monkeypatch):
    """attach(): operators written like the reference's QuantOpr (`self.weight_quantizer(self.weight)` inline)
    pick up the grouped result -- same outputs and gradients, and no per-layer forward kernel is launched"""
    from sparsebit_amd.common import Backend
    from sparsebit_amd.config import quantizer_config
    from sparsebit_amd.group import WeightQuantGroup
    from sparsebit_amd.quantizers import build_quantizer

    class QLinear(torch.nn.Module):
        def __init__(self, i, o):
            super().__init__()
            self.weight = torch.nn.Parameter(torch.randn(o, i) * 0.1)
            self.weight_quantizer = build_quantizer(quantizer_config("per-channel-symmetric", 4, quantizer="lsq"))
            self.weight_quantizer.set_backend(Backend.VIRTUAL)

        def forward(self, x):
            return torch.nn.functional.linear(x, self.weight_quantizer(self.weight))

    torch.manual_seed(1)
    model = torch.nn.Sequential(QLinear(64, 128), torch.nn.ReLU(), QLinear(128, 32), torch.nn.ReLU(), QLinear(32, 16)).cuda()
    oprs = [m for m in model if isinstance(m, QLinear)]
    for m in oprs:
        m.weight_quantizer.update_observer(m.weight.detach())
        m.weight_quantizer.calc_qparams()
        m.weight_quantizer.enable_quant()
    x = torch.randn(8, 64, device="cuda")

    def run():
        model.zero_grad(set_to_none=True)
        y = model(x)
        y.square().sum().backward()
        return y.detach().clone(), [m.weight.grad.clone() for m in oprs], [m.weight_quantizer.scale.grad.clone() for m in oprs]

    ref = run()
    group = WeightQuantGroup([(m.weight_quantizer, m.weight, None) for m in oprs])
    handles = group.attach(model)
    calls = []
    real, real_lsq = ops.fake_quant, ops.lsq_fake_quant
    monkeypatch.setattr(ops, "fake_quant", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    monkeypatch.setattr(ops, "lsq_fake_quant", lambda *a, **k: (calls.append(1), real_lsq(*a, **k))[1])
    got = run()
    assert not calls, "a member quantizer launched its own forward kernel"
    assert torch.equal(got[0], ref[0])
    for a, b in zip(got[1], ref[1]):
        assert torch.equal(a, b)
    for a, b in zip(got[2], ref[2]):
        assert torch.allclose(a, b, rtol=1e-5, atol=2e-5 * max(1.0, float(b.abs().max())))
    assert all(m.weight_quantizer._pregrouped is None for m in oprs)
    for h in handles:
        h.remove()
    got2 = run()
    assert calls and torch.equal(got2[0], ref[0])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("masked", [False, True])
def test_group_backward_equals_per_tensor_and_oracle(oracle, ops, dtype, masked):
    g = torch.Generator().manual_seed(29)
    entries, masks, gys, lsq, want, ratios = [], [], [], [], [], []
    for i, shape in enumerate(GROUP_SHAPES):
        x = (torch.randn(shape, generator=g) * (0.5 + i)).to(dtype).cuda()
        per_channel = i % 3 != 2
        qmin, qmax = [(-8, 7), (-128, 127), (0, 255), (0, 15)][i % 4]
        C = shape[0] if per_channel else 1
        scale = (torch.rand(C, generator=g) * 0.1 + 0.01).cuda() * (0.5 + i)
        if i % 2:
            scale = scale * torch.where(torch.rand(C, generator=g) > 0.5, 1.0, -1.0).cuda()  # LSQ items: signed steps
        zp = torch.zeros(C).cuda() if qmin < 0 else torch.randint(0, qmax, (C,), generator=g).float().cuda()
        entries.append((x, scale, zp, qmin, qmax))
        masks.append((torch.rand(shape, generator=g) > 0.4).cuda() if masked else None)
        gys.append(torch.randn(shape, generator=g).to(dtype).cuda())
        lsq.append(bool(i % 2))
        want.append(i != 4)
        ratios.append(0.37 if i % 2 else 1.0)
    gb = ops.GroupFakeQuantBackward(entries, masks=masks if masked else None, lsq=lsq, want_gs=want, gs_ratios=ratios)
    for rep in range(2):
        gxs, gss = gb(gys)
        for i, (x, scale, zp, qmin, qmax) in enumerate(entries):
            s_eff = scale.abs() if lsq[i] else scale
            xm = x if not masked else x * masks[i]
            gx, gs, _ = ops.fake_quant_backward(xm, gys[i], s_eff, zp, qmin, qmax, 0, True, False)
            if masked:
                gx = gx * masks[i]
            assert torch.equal(gxs[i], gx), i
            if not want[i]:
                assert gss[i] is None
                continue
            if lsq[i]:
                gs = gs * ratios[i] * torch.sign(scale)
            # two fp32/fp64 summation orders of sums with cancellation: tolerance relative to the largest entry
            assert torch.allclose(gss[i], gs, rtol=1e-5, atol=2e-5 * max(1.0, float(gs.abs().max()))), (i, (gss[i] - gs).abs().max())
    # the oracle directly on one per-channel item (fp32 only: its gy is fp32)
    if dtype == torch.float32:
        i = 3
        x, scale, zp, qmin, qmax = entries[i]
        xm = x if not masked else x * masks[i]
        ogx, ogs, _ = oracle.ste_backward(xm.cpu().numpy(), gys[i].cpu().numpy(), scale.abs().cpu().numpy(),
                                          zp.cpu().numpy(), qmin, qmax, 0)
        want_gx = torch.from_numpy(ogx).cuda() * (masks[i] if masked else 1)
        assert same_values(gxs[i].cpu().numpy(), want_gx.cpu().numpy())
        ogs = ogs * ratios[i] * np.sign(scale.cpu().numpy())
        # oracle: fp32 products summed in fp64; kernel: fp32 sums of 8, then fp64 -- sums with cancellation
        assert np.allclose(gss[i].cpu().numpy(), ogs, rtol=1e-5, atol=2e-5 * float(np.abs(ogs).max()))


def test_group_launch_rejects_unsupported(ops):
    from sparsebit_amd.lib import SbqError

    s1, z1 = torch.ones(1, device="cuda"), torch.zeros(1, device="cuda")
    with pytest.raises(SbqError):  # 147 elements per row: no whole packs (first ResNet conv) -> one by one
        ops.GroupFakeQuant([(torch.randn(64, 3, 7, 7, device="cuda"), torch.ones(64, device="cuda"),
                             torch.zeros(64, device="cuda"), -128, 127)])
    assert not ops.GroupFakeQuant.supports(torch.randn(64, 3, 7, 7, device="cuda"))
    assert ops.GroupFakeQuant.supports(torch.randn(64, 64, 3, 3, device="cuda"))
    with pytest.raises(SbqError):  # CPU tensor: no fallback
        ops.GroupFakeQuant([(torch.randn(8, 8), s1, z1, -8, 7)])
    with pytest.raises(SbqError):
        ops.GroupFakeQuant([])


@pytest.mark.parametrize("shape", [(11008, 4096), (16384, 4096), (9000, 8200), (3, 4096 * 4096 + 8)])
def test_large_grid_paths(oracle, ops, shape):
    """grids beyond one resident wave: several tiles per workgroup (software-pipelined loop), the
    two-tiles-per-iteration variant (>= 32768 slabs), ragged last slabs, very long rows.  Checked by
    (1) dq == q * scale, (2) the oracle on sampled rows incl. the first / last tiles, (3) every
    forced launch variant producing the same bits as the automatic one."""
    from sparsebit_amd import lib as L

    g = torch.Generator().manual_seed(shape[0])
    C, inner = shape
    x = (torch.randn(C, 1, generator=g) * torch.randn(1, 4099, generator=g)).repeat(1, inner // 4099 + 1)[:, :inner]
    x = (x * torch.logspace(-1, 1, C).unsqueeze(1)).bfloat16().cuda().contiguous()
    mn, mx, _ = ops.channel_stats(x, 0, True)
    scale, zp = ops.qparams_from_minmax(mn, mx, -128, 127, True)
    y, qi = ops.fake_quant(x, scale, zp, -128, 127, 0, return_q=torch.int8)
    assert torch.equal(y, qi.float() * scale.reshape(-1, 1))
    rows = sorted({0, 1, C // 2, C - 2, C - 1})
    xr = x[rows].float().cpu().numpy()
    ref_dq, ref_q = oracle.qdq(xr, scale[rows].cpu().numpy(), zp[rows].cpu().numpy(), -128, 127, 0)
    assert same_values(y[rows].cpu().numpy(), ref_dq)
    assert np.array_equal(qi[rows].cpu().numpy().astype(np.int32), ref_q)
    y16 = ops.fake_quant(x, scale, zp, -128, 127, 0, out_dtype=torch.bfloat16)
    assert torch.equal(y16, y.bfloat16())
    try:
        for variant in (0, 1, 2, 4, 5):  # U = 1 / 2 / 4, cached instead of nontemporal
            L.set_tuning(0, variant)
            assert torch.equal(ops.fake_quant(x, scale, zp, -128, 127, 0, out_dtype=torch.bfloat16), y16), variant
        L.set_tuning(0, -1)
        for cap in (256, 1000):  # many tiles per workgroup
            L.set_tuning(1, cap)
            assert torch.equal(ops.fake_quant(x, scale, zp, -128, 127, 0, out_dtype=torch.bfloat16), y16), cap
    finally:
        L.set_tuning(0, -1)
        L.set_tuning(1, 0)


# --------------------------------------------------------------------------------------
# observers on random data vs the oracle
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape,ch_axis", SHAPES + [((3, 8192 + 24), 0), ((1, 70001), 0), ((64, 197, 384), 2),
                                                     ((7, 3000, 8), 2)])
@pytest.mark.parametrize("dtype", DTYPES)
def test_channel_stats_vs_oracle(oracle, ops, shape, ch_axis, dtype):
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(*shape, generator=g) * 2).to(dtype)
    xf = x.float().numpy()
    for perch in (True, False):
        mn, mx, ab = ops.channel_stats(x.cuda(), ch_axis, perch, want_abssum=True)
        rmn, rmx = oracle.minmax(xf, ch_axis, perch)
        assert np.array_equal(mn.cpu().numpy(), rmn) and np.array_equal(mx.cpu().numpy(), rmx)
        axes = tuple(i for i in range(x.ndim) if not (perch and i == ch_axis))
        want = np.abs(xf.astype(np.float64)).sum(axis=axes).reshape(-1)
        assert np.allclose(ab.cpu().numpy(), want, rtol=1e-6)
    # NaN propagates like torch.min / torch.max
    xn = x.clone().float()
    xn.view(-1)[xn.numel() // 2] = float("nan")
    mn, mx, _ = ops.channel_stats(xn.cuda(), ch_axis, False)
    assert np.isnan(mn.item()) and np.isnan(mx.item())


@pytest.mark.parametrize("sym", [True, False])
def test_qparams_kernel_vs_oracle(oracle, ops, sym):
    g = np.random.default_rng(3)
    mn = np.concatenate([g.normal(size=500).astype(np.float32) * 3, [0, 1, -1, 0, np.nan, -np.inf, 1e-30]]).astype(np.float32)
    mx = np.concatenate([g.normal(size=500).astype(np.float32) * 3, [0, 3, 3, 1e-9, 1.0, np.inf, 2e-30]]).astype(np.float32)
    for qmin, qmax in ((-128, 127), (0, 255), (-8, 7), (0, 15), (-32768, 32767)):
        s, z = ops.qparams_from_minmax(dev_tensor(mn), dev_tensor(mx), qmin, qmax, sym)
        rs, rz = oracle.qparams_from_minmax(mn, mx, qmin, qmax, sym)
        assert same_values(s.cpu().numpy(), rs) and same_values(z.cpu().numpy(), rz)


@pytest.mark.parametrize("shape,ch_axis,perch", [((64, 96), 0, True), ((32, 16, 3, 3), 0, True), ((24, 3, 7, 7), 0, True),
                                                  ((5, 4096 + 8), 0, True), ((4, 8, 6, 6), 1, True), ((3, 17, 48), 2, False),
                                                  ((2, 33333), 0, False)])
@pytest.mark.parametrize("scheme", [("sym", -128, 127), ("aff", 0, 255), ("sym", -8, 7)])
def test_mse_observer_vs_oracle(oracle, ops, shape, ch_axis, perch, scheme):
    """Per-channel MSE follows the CUDA-kernel semantics.  Candidate index must
    equal the oracle's unless the two candidates' fp64 losses agree to 1e-7 relative (fp32
    summation order is free; the reference's own order is unspecified)."""
    kind, qmin, qmax = scheme
    g = torch.Generator().manual_seed(21)
    x = (torch.randn(*shape, generator=g) * 1.7).bfloat16()
    xf = x.float().numpy()
    from sparsebit_amd.config import quantizer_config
    from sparsebit_amd.observers import build_observer
    from sparsebit_amd.quantizers.quant_descriptor import QuantDescriptor

    bit = {127: 8, 255: 8, 7: 4}[qmax]
    name = "per-%s-%s" % ("channel" if perch else "tensor", "symmetric" if kind == "sym" else "affine")
    layout = {0: None, 1: "NCHW", 2: "NLC"}[ch_axis]
    cfg = quantizer_config(name, bit, observer="MSE", target="weight" if layout is None else "feature", layout=layout or "NCHW")
    obs = build_observer(cfg, QuantDescriptor(cfg))
    obs.data_cache.update(x.cuda())
    s, z = obs.calc_qparams()
    rs, rz, rbest, rsse = oracle.mse(xf, qmin, qmax, kind == "sym", ch_axis, perch)
    best = obs.best_index.cpu().numpy()
    for c in np.nonzero(best != rbest)[0]:
        a, b = rsse[c, best[c]], rsse[c, rbest[c]]
        assert abs(a - b) <= 1e-7 * max(a, b), (c, best[c], rbest[c], a, b)
    same = best == rbest
    assert same.mean() > 0.97
    assert np.array_equal(s.reshape(-1).cpu().numpy()[same], rs[same])
    assert same_values(z.reshape(-1).cpu().numpy()[same], rz[same])


@pytest.mark.parametrize("shape", [(64, 96), (16, 4096), (7, 16384), (33, 147), (5, 1)])
@pytest.mark.parametrize("alpha", [1e-3, 0.05, 0.5, 0.0, 1.0])
@pytest.mark.parametrize("dtype", DTYPES)
def test_percentile_rows_vs_oracle(oracle, ops, shape, alpha, dtype):
    g = torch.Generator().manual_seed(31)
    x = (torch.randn(*shape, generator=g)).to(dtype)
    x[0] = x[0].abs()  # a row without negatives -> min stays 0
    if shape[0] > 2:
        x[1] = -x[1].abs() - 1  # a row without non-negatives -> max stays 0
    x.view(-1)[::7] = x.view(-1)[0]  # many duplicates
    mn, mx = ops.percentile_rows(x.cuda(), alpha)
    rmn, rmx = oracle.percentile(x.float().numpy(), alpha, 0, True)
    assert same_values(mn.cpu().numpy(), rmn) and same_values(mx.cpu().numpy(), rmx)


@pytest.mark.parametrize("case", [((3, 50, 64), 2, False), ((4, 8, 6, 6), 1, True), ((2, 70001), 0, False),
                                  ((6, 20000), 0, True), ((3, 17, 48), 2, True)])
@pytest.mark.parametrize("alpha", [1e-3, 0.01, 0.3])
def test_percentile_radix_path_vs_oracle(oracle, case, alpha):
    """The general (shardable) radix path: several cached batches, per tensor and per channel."""
    shape, ch_axis, perch = case
    from sparsebit_amd.config import quantizer_config
    from sparsebit_amd.observers import build_observer
    from sparsebit_amd.quantizers.quant_descriptor import QuantDescriptor

    g = torch.Generator().manual_seed(41)
    xs = [torch.randn(*shape, generator=g).bfloat16() for _ in range(3)]
    layout = {0: None, 1: "NCHW", 2: "NLC"}[ch_axis]
    name = "per-%s-affine" % ("channel" if perch else "tensor")
    cfg = quantizer_config(name, 8, observer="PERCENTILE", target="weight" if layout is None else "feature",
                           layout=layout or "NCHW", alpha=alpha)
    obs = build_observer(cfg, QuantDescriptor(cfg))
    use = xs if layout is not None else xs[:1]
    for x in use:
        obs.data_cache.update(x.cuda())
    mn, mx = obs.calc_minmax()
    if perch:
        # per channel over the union of batches (documented deviation from reference quirk Q3)
        data = np.concatenate([np.moveaxis(x.float().numpy(), ch_axis, 0).reshape(x.shape[ch_axis], -1) for x in use], 1)
        rmn, rmx = oracle.percentile(data, alpha, 0, True)
    else:
        data = np.concatenate([x.float().numpy().reshape(-1) for x in use])
        rmn, rmx = oracle.percentile(data, alpha, per_channel=False)
    assert same_values(mn.reshape(-1).cpu().numpy(), rmn) and same_values(mx.reshape(-1).cpu().numpy(), rmx)


# --------------------------------------------------------------------------------------
# unstructured mask, fused mask + QDQ
# --------------------------------------------------------------------------------------
def test_percentile_ranks_from_first_histogram(ops):
    """sbq_percentile_ranks: neg / pos counts (with -0.0 counted as >= 0 and NaN as neither, like
    percentile.py:27-28) and Python-rounded ranks, all from the pass-0 histogram"""
    g = torch.Generator().manual_seed(11)
    x = torch.randn(5, 3001, generator=g)
    x[0, :7] = -0.0
    x[1, 5:9] = float("nan")
    x[2] = x[2].abs()  # no negatives
    x[3] = -x[3].abs() - 1  # no non-negatives
    x[4, ::2] = 0.0
    xd = x.cuda()
    be = ops.HipSelectBackend()
    for alpha in (1e-3, 0.0125, 0.5):
        state = be.zero_state(5, 2, xd.device)
        hist = be.new_hist(5, 2, xd.device)
        be.histogram(xd, state, hist, 0, 2, False, 0, True)
        counts = be.percentile_ranks(hist, state, alpha, 5).cpu()
        neg, pos = (x < 0).sum(1), (x >= 0).sum(1)
        assert torch.equal(counts[0], neg) and torch.equal(counts[1], pos)
        n = x.shape[1]
        for c in range(5):
            k_min = min(max(max(round(int(neg[c]) * alpha), 1), 1), n)
            k_max = min(max(n - max(round(int(pos[c]) * alpha), 0), 1), n)
            assert state[c, 0].tolist() == [0, k_min] and state[c, 1].tolist() == [0, k_max], (alpha, c)


@pytest.mark.parametrize("ratio", [0.0, 0.3, 0.5, 0.9, 1.0])
@pytest.mark.parametrize("wname", ["lin", "conv"])
def test_l1_mask_vs_reference_golden(golden, ratio, wname):
    from sparsebit_amd.config import sparser_config
    from sparsebit_amd.sparsers import build_sparser

    x = golden["uni/per-channel-symmetric/8/%s/x" % wname]
    sp = build_sparser(sparser_config(ratio))
    m = sp.calc_mask(dev_tensor(x))
    assert np.array_equal(m.cpu().numpy().astype(np.uint8), golden["mask/%g/%s" % (ratio, wname)])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n", [8, 1000, 4097, 300001])
def test_l1_mask_random_vs_oracle(oracle, dtype, n):
    from sparsebit_amd.config import sparser_config
    from sparsebit_amd.sparsers import build_sparser

    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, generator=g).to(dtype)
    x[::5] = x[0]  # ties with a repeated magnitude
    for ratio in (0.1, 0.5, 0.77):
        sp = build_sparser(sparser_config(ratio))
        m = sp.calc_mask(x.cuda())
        rm, rt = oracle.l1_mask(x.float().numpy(), ratio)
        assert float(sp.calc_threshold(x.cuda())) == float(rt)
        assert np.array_equal(m.cpu().numpy(), rm)
        assert m.dtype == torch.bool


def test_mask_kat_and_fused_mask_lsq(golden, oracle, ops):
    from sparsebit_amd.common import Backend
    from sparsebit_amd.config import quantizer_config, sparser_config
    from sparsebit_amd.quantizers import build_quantizer
    from sparsebit_amd.sparsers import build_sparser

    sp = build_sparser(sparser_config(0.5))
    m = sp.calc_mask(dev_tensor(golden["mask/kat_x"]))
    assert m.int().cpu().tolist() == [[0, 0, 0, 1], [0, 1, 1, 0]]
    # config 5's composition: LSQ 4-bit weight quantizer applied to weight * mask
    x = dev_tensor(golden["maskq/x"])
    mask = sp.calc_mask(x)
    assert np.array_equal(mask.cpu().numpy().astype(np.uint8), golden["maskq/mask"])
    q = build_quantizer(quantizer_config("per-channel-symmetric", 4, quantizer="lsq"))
    q.set_backend(Backend.VIRTUAL)
    q.update_observer(x)
    q.calc_qparams()
    q.enable_quant()
    # use the reference's own scale so that the comparison is bit-exact
    with torch.no_grad():
        q.scale.copy_(dev_tensor(golden["maskq/scale"]).reshape(q.scale.shape))
    want = golden["maskq/dq"]
    assert same_values(q(x * mask).detach().cpu().numpy(), want)  # unfused: mask multiply then quantizer
    assert same_values(q.forward_masked(x, mask=mask).cpu().numpy(), want)  # fused, mask bytes
    assert same_values(q.forward_masked(x, thresh=sp.calc_threshold(x)).cpu().numpy(), want)  # fused, threshold
    y16 = q.forward_masked(x.bfloat16(), mask=mask, out_dtype=torch.bfloat16)
    assert same_values(y16.float().cpu().numpy(), torch.from_numpy(want).bfloat16().float().numpy())


# --------------------------------------------------------------------------------------
# STE / LSQ backward
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["bwd/pc4", "bwd/pt8a", "bwd/pc8a_nchw"])
def test_ste_backward_vs_reference_golden(golden, oracle, ops, name):
    qmin, qmax, ch_axis = [int(v) for v in golden[name + "/meta"]]
    x, gy = golden[name + "/x"], golden[name + "/gy"]
    s, z = golden[name + "/scale"], golden[name + "/zero_point"]
    gx, gs, gz = ops.fake_quant_backward(dev_tensor(x), dev_tensor(gy), dev_tensor(s), dev_tensor(z), qmin, qmax, ch_axis)
    assert same_values(gx.cpu().numpy(), golden[name + "/gx"])
    assert np.allclose(gs.cpu().numpy(), golden[name + "/gs"], rtol=1e-5, atol=1e-5)
    assert np.allclose(gz.cpu().numpy(), golden[name + "/gzp"], rtol=1e-5, atol=1e-5)
    ogx, ogs, ogz = oracle.ste_backward(x, gy, s, z, qmin, qmax, ch_axis)
    assert same_values(gx.cpu().numpy(), ogx)
    assert np.allclose(gs.cpu().numpy(), ogs, rtol=1e-6, atol=1e-6)
    assert np.allclose(gz.cpu().numpy(), ogz, rtol=1e-6, atol=1e-6)


def test_lsq_training_step_autograd(oracle):
    """config 5 shape of use: LSQ 4-bit per-channel weight quantizer inside autograd."""
    from sparsebit_amd.common import Backend
    from sparsebit_amd.config import quantizer_config
    from sparsebit_amd.quantizers import build_quantizer
    import math

    g = torch.Generator().manual_seed(9)
    w = torch.randn(48, 16, 3, 3, generator=g).cuda().requires_grad_(True)
    q = build_quantizer(quantizer_config("per-channel-symmetric", 4, quantizer="lsq"))
    q.set_backend(Backend.VIRTUAL)
    q.update_observer(w)
    q.calc_qparams()
    q.enable_quant()
    out = q(w)
    gy = torch.randn(out.shape, generator=g).cuda()
    out.backward(gy)
    s = q.scale.detach().abs().reshape(-1).cpu().numpy()
    gx, gs, gz = oracle.ste_backward(w.detach().cpu().numpy(), gy.cpu().numpy(), s, np.zeros_like(s), -8, 7, 0)
    assert same_values(w.grad.cpu().numpy(), gx)
    ratio = 1.0 / math.sqrt(16 * 9 * 7)  # lsq.py:70-71
    sign = np.sign(q.scale.detach().reshape(-1).cpu().numpy())
    assert np.allclose(q.scale.grad.reshape(-1).cpu().numpy(), gs * ratio * sign, rtol=1e-5, atol=1e-7)


# --------------------------------------------------------------------------------------
# GPTQ 4-bit mat-vec
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["g128", "g-1", "rag"])
def test_gptq_vs_reference_golden(golden, oracle, name):
    from sparsebit_amd import gptq

    B, M, N, GS = [int(v) for v in golden["gptq/%s/meta" % name]]
    layer = torch.nn.Linear(M, N).cuda()
    with torch.no_grad():
        layer.weight.copy_(dev_tensor(golden["gptq/%s/w" % name]))
        layer.bias.copy_(dev_tensor(golden["gptq/%s/bias" % name]))
    qz = gptq.Quantizer()
    qz.configure(bit=4, perchannel=True, sym=False, mse=False)
    qz.find_params(layer.weight.data, weight=True, groupsize=GS)
    assert np.array_equal(qz.scale.reshape(N, -1).cpu().numpy(), golden["gptq/%s/scale" % name])
    assert np.array_equal(qz.zero.reshape(N, -1).cpu().numpy(), golden["gptq/%s/zero" % name])
    layer.weight.data = gptq.quantize(layer.weight.data.view(-1, M if GS == -1 else GS), qz.scale.view(-1, 1),
                                      qz.zero.view(-1, 1), qz.maxq).view(N, M)
    assert np.array_equal(layer.weight.data.cpu().numpy(), golden["gptq/%s/wq" % name])
    ql = gptq.QuantLinear(M, N, bit=4, groupsize=GS)
    ql.pack(layer, qz.scale, qz.zero)
    assert np.array_equal(ql.qweight.cpu().numpy(), golden["gptq/%s/qweight" % name])
    y = ql(dev_tensor(golden["gptq/%s/x" % name]))
    # the reference's own tolerance (test_cuda_kernel.py:45)
    assert np.allclose(y.cpu().numpy(), golden["gptq/%s/y" % name], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,M,N,GS", [(1, 4096, 4096, 128), (2, 4096, 4096, 128), (1, 11008, 4096, 128), (2, 1000, 96, -1),
                                       (8, 4096, 4096, 128), (32, 1024, 1280, 128), (4, 6661, 1027, -1),
                                       (2, 4096, 11008, 128), (1, 1024, 256, 256), (3, 4096, 4096, 128), (4, 11008, 4096, 128),
                                       (4, 1024, 256, -1), (3, 1000, 96, -1)])
def test_gptq_random_vs_oracle(oracle, B, M, N, GS):
    """shapes after test_cuda_kernel.py:48-126 (incl. an irregular M, N and group sizes)"""
    from sparsebit_amd import gptq

    torch.manual_seed(B * 7 + M)
    layer = torch.nn.Linear(M, N)
    x = torch.randn(B, M)
    w = layer.weight.data.numpy()
    scale, zero = oracle.gptq_find_params(w, 4, GS)
    wq = oracle.gptq_quantize(w, scale, zero)
    qw, zeros_p = oracle.gptq_pack4(wq, scale, zero)
    ql = gptq.QuantLinear(M, N, bit=4, groupsize=GS)
    ql.qweight = torch.from_numpy(qw)
    ql.scales = torch.from_numpy(scale).reshape(ql.scales.shape)
    ql.zeros = torch.from_numpy(zeros_p).reshape(ql.zeros.shape)
    ql.bias = layer.bias.detach().clone()
    ql = ql.cuda()
    y = ql(x.cuda())
    want = x.double() @ torch.from_numpy(wq).double().t() + layer.bias.detach().double()
    assert torch.allclose(y.double().cpu(), want, rtol=1e-5, atol=1e-5)
    y2 = ql(x.cuda())
    assert torch.equal(y, y2)  # deterministic: no float atomics
    if M * N <= 4096 * 1280:
        ref = oracle.vecquant4matmul(x.numpy(), qw, layer.bias.detach().numpy(), scale, zeros_p, GS)
        assert np.allclose(y.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", ["b3/g128", "b3/g-1", "b3/rag", "b3/strip", "b2/g64", "b2/g-1", "b2/rag", "b2/strip"])
def test_gptq_low_bit_vs_reference_golden(golden, oracle, name):
    from sparsebit_amd import gptq

    B, M, N, GS, bit = [int(v) for v in golden["gptq/%s/meta" % name]]
    layer = torch.nn.Linear(M, N).cuda()
    with torch.no_grad():
        layer.weight.copy_(dev_tensor(golden["gptq/%s/w" % name]))
        layer.bias.copy_(dev_tensor(golden["gptq/%s/bias" % name]))
    qz = gptq.Quantizer()
    qz.configure(bit=bit, perchannel=True, sym=False, mse=False)
    qz.find_params(layer.weight.data, weight=True, groupsize=GS)
    assert np.array_equal(qz.scale.reshape(N, -1).cpu().numpy(), golden["gptq/%s/scale" % name])
    assert np.array_equal(qz.zero.reshape(N, -1).cpu().numpy(), golden["gptq/%s/zero" % name])
    layer.weight.data = gptq.quantize(layer.weight.data.view(-1, M if GS == -1 else GS), qz.scale.view(-1, 1),
                                      qz.zero.view(-1, 1), qz.maxq).view(N, M)
    assert np.array_equal(layer.weight.data.cpu().numpy(), golden["gptq/%s/wq" % name])
    ql = gptq.QuantLinear(M, N, bit=bit, groupsize=GS)
    ql.pack(layer, qz.scale, qz.zero)
    assert np.array_equal(ql.qweight.cpu().numpy(), golden["gptq/%s/qweight" % name])
    y = ql(dev_tensor(golden["gptq/%s/x" % name]))
    assert np.allclose(y.cpu().numpy(), golden["gptq/%s/y" % name], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("bit", [3, 2])
@pytest.mark.parametrize("B,M,N,GS", [(1, 4096, 4096, 128), (2, 4096, 4096, 128), (1, 11008, 4096, 128), (2, 1000, 96, -1),
                                       (8, 4096, 4096, 128), (32, 1024, 1280, 128), (4, 6661, 1027, -1),
                                       (1, 1024, 256, 256), (3, 1024, 256, 64), (1, 1024, 256, 64), (4, 4096, 4096, 128),
                                       (3, 11008, 4096, 128)])
def test_gptq_low_bit_random_vs_oracle(oracle, bit, B, M, N, GS):
    from sparsebit_amd import gptq

    if bit == 3 and GS == 64:
        pytest.skip("3-bit groups are multiples of 128 (cuda_kernel_3bit.cu:56-59)")
    torch.manual_seed(B * 7 + M + bit)
    layer = torch.nn.Linear(M, N)
    x = torch.randn(B, M)
    w = layer.weight.data.numpy()
    scale, zero = oracle.gptq_find_params(w, bit, GS)
    wq = oracle.gptq_quantize(w, scale, zero, bit)
    qw, zeros_p = oracle.gptq_pack(wq, scale, zero, bit)
    ql = gptq.QuantLinear(M, N, bit=bit, groupsize=GS)
    ql.qweight = torch.from_numpy(qw)
    ql.scales = torch.from_numpy(scale).reshape(ql.scales.shape)
    ql.zeros = torch.from_numpy(zeros_p).reshape(ql.zeros.shape)
    ql.bias = layer.bias.detach().clone()
    ql = ql.cuda()
    # device-side pack of the same quantized weight gives the same words
    lay_q = torch.nn.Linear(M, N).cuda()
    with torch.no_grad():
        lay_q.weight.copy_(torch.from_numpy(wq))
    ql2 = gptq.QuantLinear(M, N, bit=bit, groupsize=GS).cuda()
    ql2.pack(lay_q, torch.from_numpy(scale).reshape(ql.scales.shape).cuda(), torch.from_numpy(zero).reshape(ql.scales.shape).cuda())
    assert torch.equal(ql2.qweight, ql.qweight)
    y = ql(x.cuda())
    want = x.double() @ torch.from_numpy(wq).double().t() + layer.bias.detach().double()
    assert torch.allclose(y.double().cpu(), want, rtol=1e-5, atol=2e-5)
    assert torch.equal(y, ql(x.cuda()))  # deterministic
    if M * N <= 4096 * 1280:
        ref = oracle.vecquantmatmul(x.numpy(), qw, layer.bias.detach().numpy(), scale, zeros_p, GS, bit)
        assert np.allclose(y.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)


def test_gptq_kernel_module_entry_points(golden):
    """the six names of the reference's pybind module, reference argument order, in-place accumulate"""
    from sparsebit_amd import gptq

    for name, bit in (("g128", 4), ("b3/g128", 3), ("b2/g64", 2), ("g-1", 4), ("b3/g-1", 3), ("b2/g-1", 2)):
        meta = [int(v) for v in golden["gptq/%s/meta" % name]]
        B, M, N, GS = meta[:4]
        x = dev_tensor(golden["gptq/%s/x" % name])
        qw = torch.from_numpy(golden["gptq/%s/qweight" % name]).cuda()
        sc = dev_tensor(golden["gptq/%s/scales" % name])
        zr = dev_tensor(golden["gptq/%s/zeros" % name])
        y = dev_tensor(golden["gptq/%s/bias" % name]).repeat(B, 1).contiguous()
        if GS == -1:
            getattr(gptq.cuda_kernel, "vecquant%dmatmul" % bit)(x, qw, y, sc, zr)
        else:
            getattr(gptq.cuda_kernel, "vecgroupquant%dmatmul" % bit)(x, qw, y, sc, zr, GS)
        assert np.allclose(y.cpu().numpy(), golden["gptq/%s/y" % name], rtol=1e-5, atol=1e-5)


def test_gptq_rejects_wrong_packing():
    from sparsebit_amd import ops
    from sparsebit_amd.lib import SbqError

    x = torch.zeros(1, 256, device="cuda")
    out = torch.zeros(1, 32, device="cuda")
    sc = torch.ones(32, 1, device="cuda")
    with pytest.raises(SbqError):  # 4-bit sized buffer handed to the 3-bit kernel
        ops.vecquantmatmul(3, x, torch.zeros(32, 32, dtype=torch.int32, device="cuda"), out, sc, sc, 0)
    with pytest.raises(SbqError):
        ops.vecquantmatmul(5, x, torch.zeros(32, 32, dtype=torch.int32, device="cuda"), out, sc, sc, 0)


# --------------------------------------------------------------------------------------
# real quantized storage
# --------------------------------------------------------------------------------------
EXPORT_CASES = [c for c in QUANT_CASES if c.split("/")[0] in ("uni", "act", "pct", "mse") and "/16/" not in c]


@pytest.mark.parametrize("name", EXPORT_CASES)
def test_qdq_export_matches_reference_fake_quant(golden, oracle, name):
    """QuantizeLinear constants from the kernel: DequantizeLinear(q) == the reference's fake-quant
    output bit for bit, q == round(x/s)+zp saturated (ONNX QuantizeLinear), containers as in
    torch_fake_quant (int8 / uint8)."""
    from sparsebit_amd import export
    from sparsebit_amd.quantizers import build_quantizer

    cfg, backend = case_config(name)
    qmin, qmax, ch_axis, perch, sym = _meta(golden, name)
    if qmax - qmin > 255:
        pytest.skip("wider than the 8-bit QDQ container")
    q = build_quantizer(cfg)
    q.set_backend(backend)
    xs = all_x(golden, name)
    for x in xs:
        q.update_observer(dev_tensor(x))
    q.calc_qparams()
    x0 = dev_tensor(xs[0])
    dq, rec = export.quantize_linear(q, x0)
    assert same_values(dq.cpu().numpy(), golden[name + "/dq"])
    assert same_values(rec.dequantize().cpu().numpy(), golden[name + "/dq"])
    want_dtype = torch.int8 if qmin < 0 else torch.uint8
    assert rec.q.dtype == want_dtype and rec.zero_point.dtype == want_dtype and rec.scale.dtype == torch.float32
    assert rec.bits == q.bit and rec.axis == (ch_axis if perch else None)
    _, ref_q = oracle.qdq(xs[0], golden[name + "/scale"], golden[name + "/zero_point"], qmin, qmax, ch_axis)
    assert np.array_equal(rec.levels().cpu().numpy().astype(np.int32), ref_q)
    assert int(rec.levels().min()) >= qmin and int(rec.levels().max()) <= qmax
    back = export.QDQTensor.from_state_dict({k: v.cpu() for k, v in rec.state_dict("w.").items()}, "w.").to("cuda")
    assert same_values(back.dequantize().cpu().numpy(), golden[name + "/dq"])
    none_dq, rec_q = export.quantize_linear(q, x0, dequantized=False)  # QuantizeLinear alone: same levels
    assert none_dq is None and torch.equal(rec_q.q, rec.q)
    assert torch.equal(rec.dequantize(torch.bfloat16), dq.bfloat16())
    if qmax - qmin <= 15 and xs[0].shape[-1] % 8 == 0 and not (perch and ch_axis == xs[0].ndim - 1):
        dq4, rec4 = export.quantize_linear(q, x0, pack_int4=True)
        assert rec4.packed and rec4.q.dtype == torch.uint8 and rec4.q.numel() * 2 == x0.numel()
        assert torch.equal(rec4.levels(), rec.levels())
        assert torch.equal(rec4.q, export.pack_int4(rec.levels()))
        assert same_values(rec4.dequantize().cpu().numpy(), golden[name + "/dq"])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape,ch_axis,lo,hi", [((4096, 4096), 0, -8, 7), ((512, 512, 3, 3), 0, 0, 15),
                                                 ((64, 197, 384), 2, -8, 7), ((3, 1000, 64), None, 0, 15),
                                                 ((33, 72), 0, -2, 1)])
def test_packed_int4_output_equals_packed_levels(ops, dtype, shape, ch_axis, lo, hi):
    """SBQ_Q_I4 straight from the kernel == pack_int4 of the int8 levels, all layouts the pack kernels cover"""
    from sparsebit_amd import export

    g = torch.Generator().manual_seed(5)
    x = (torch.randn(shape, generator=g) * 3).to(dtype).cuda()
    C = 1 if ch_axis is None else shape[ch_axis]
    scale = (torch.rand(C, generator=g) * 0.5 + 0.25).cuda()
    zp = torch.full((C,), 0.0 if lo < 0 else 7.0).cuda()
    ax = 0 if ch_axis is None else ch_axis
    y8, q8 = ops.fake_quant(x, scale, zp, lo, hi, ax, return_q=torch.int8 if lo < 0 else torch.uint8)
    y4, q4 = ops.fake_quant(x, scale, zp, lo, hi, ax, return_q="int4")
    assert torch.equal(y4, y8)
    assert q4.dtype == torch.uint8 and q4.numel() * 2 == x.numel()
    assert torch.equal(q4, export.pack_int4(q8))
    assert torch.equal(export.unpack_int4(q4, lo < 0).reshape(shape), q8)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape,ch_axis", SHAPES + [((4096, 4096), 0)])
def test_quantize_only_and_dequantize_kernel(ops, dtype, shape, ch_axis):
    """quantize-only (y == NULL) gives the fake-quant launch's levels; sbq_dequantize_linear of them gives its
    dequantized tensor bit for bit -- int8, uint8, int32 and packed int4, every geometry path"""
    from sparsebit_amd import export

    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(shape, generator=g) * 2).to(dtype).cuda()
    for per_channel in (True, False):
        C = shape[ch_axis] if per_channel else 1
        scale = (torch.rand(C, generator=g) * 0.2 + 0.05).cuda()
        for lo, hi, zpv, rq in ((-128, 127, 0.0, torch.int8), (0, 255, 101.0, torch.uint8), (-8, 7, 0.0, torch.int32),
                                (-8, 7, 0.0, "int4"), (0, 15, 9.0, "int4")):
            zp = torch.full((C,), zpv).cuda()
            inner = int(np.prod(shape[ch_axis + 1:])) if per_channel else x.numel()
            if rq == "int4" and (inner % 8 or (per_channel and ch_axis == len(shape) - 1 and shape[-1] % 8)):
                continue  # packed int4 is emitted by the pack kernels only
            y, q = ops.fake_quant(x, scale, zp, lo, hi, ch_axis, return_q=rq)
            q2 = ops.quantize_only(x, scale, zp, lo, hi, ch_axis, return_q=rq)
            assert torch.equal(q2, q), (per_channel, lo, hi)
            packed = rq == "int4"
            back = ops.dequantize_linear(q, scale, zp, shape=shape if packed else None, ch_axis=ch_axis,
                                         signed=lo < 0, packed_int4=packed)
            assert torch.equal(back, y), (per_channel, lo, hi, rq)
            back16 = ops.dequantize_linear(q, scale, zp, shape=shape if packed else None, ch_axis=ch_axis,
                                           signed=lo < 0, packed_int4=packed, out_dtype=torch.bfloat16)
            assert torch.equal(back16, y.bfloat16())


def test_packed_int4_rejects_what_it_cannot_pack(ops):
    from sparsebit_amd.lib import SbqError

    s, z = torch.ones(1, device="cuda"), torch.zeros(1, device="cuda")
    with pytest.raises(SbqError):  # 8-bit range
        ops.fake_quant(torch.randn(64, 64, device="cuda"), s, z, -128, 127, 0, return_q="int4")
    with pytest.raises(SbqError):  # ragged tail: two elements share a byte, no scalar path
        ops.fake_quant(torch.randn(10, 10, device="cuda"), s, z, -8, 7, 0, return_q="int4")
    with pytest.raises(SbqError):  # fused mask variants are not instantiated for int4 output
        ops.fake_quant(torch.randn(64, 64, device="cuda"), s, z, -8, 7, 0, return_q="int4",
                       mask=torch.ones(64, 64, dtype=torch.bool, device="cuda"))


def test_pack8_checkpoint_layout_on_device(golden):
    from sparsebit_amd import export

    q32 = torch.from_numpy(golden["pack8/qweight32"]).cuda()
    q8 = export.pack32_to_pack8(q32)
    assert np.array_equal(q8.cpu().numpy(), golden["pack8/qweight8"])
    assert torch.equal(export.pack8_to_pack32(q8), q32)


# --------------------------------------------------------------------------------------
# widened set: remaining observers / quantizers on the same kernels
# --------------------------------------------------------------------------------------
def _cases2():
    import os

    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_golden.npz"))
    return z["cases2"].tolist()


def _cfg2(name):
    from sparsebit_amd.config import quantizer_config

    parts = name.split("/")
    kind = parts[0]
    tail = parts[-1]
    is_act = tail.startswith(("nchw", "nlc", "relu"))
    layout = "NLC" if tail.startswith("nlc") else "NCHW"
    target = "feature" if is_act else "weight"
    if kind == "ma":
        return quantizer_config(parts[1], 8, observer="MOVING_AVERAGE", target=target, layout=layout,
                                ema_ratio=0.7 if layout == "NLC" else 0.9)
    if kind == "aciq":
        return quantizer_config(parts[2], int(parts[3]), observer="ACIQ", target=target, layout=layout, aciq=parts[1])
    if kind == "pact":
        return quantizer_config(parts[1], int(parts[2]), quantizer="pact", target=target, layout=layout,
                                pact_alpha=2.0 if tail == "relu" else 1.5)
    if kind == "dorefa":
        return quantizer_config(parts[1], 4, quantizer="dorefa")
    if kind == "lsqp":
        return quantizer_config(parts[1], int(parts[2]), quantizer="lsq+", target=target, layout=layout)
    raise KeyError(name)


@pytest.mark.parametrize("name", _cases2())
def test_widened_quantizers_vs_reference_golden(golden, oracle, name):
    from sparsebit_amd.common import Backend
    from sparsebit_amd.quantizers import build_quantizer

    kind = name.split("/")[0]
    qmin, qmax, ch_axis, perch, sym = _meta(golden, name)
    q = build_quantizer(_cfg2(name))
    q.set_backend(Backend.VIRTUAL)
    xs = all_x(golden, name)
    for x in xs:
        q.update_observer(dev_tensor(x))
    scale, zp = q.calc_qparams()
    assert q.qdesc.qmin == qmin and q.qdesc.qmax == qmax and q.qdesc.is_symmetric == sym
    s = scale.detach().reshape(-1).cpu().numpy()
    z = zp.detach().reshape(-1).cpu().numpy()
    gs, gz = golden[name + "/scale"], golden[name + "/zero_point"]
    # moving average, ACIQ-gaus and PACT are the reference's fp32 operations in the same order:
    # exact.  ACIQ-laplace / LSQ+ weights sum in fp64 where torch sums fp32, DoReFa applies the
    # GPU's tanh: the contract's 1e-6 (tanh: 1e-5) relative on the parameters.
    exact = kind in ("ma", "pact") or (kind == "aciq" and "GAUS" in name) or (kind == "lsqp" and not perch)
    if exact:
        assert np.array_equal(s, gs), (s, gs)
        assert same_values(z, gz)
        assert same_values(q.observer.min_val.reshape(-1).cpu().numpy(), golden[name + "/min_val"])
        assert same_values(q.observer.max_val.reshape(-1).cpu().numpy(), golden[name + "/max_val"])
    else:
        tol = 1e-5 if kind == "dorefa" else 1e-6
        assert rel_err(s, gs) <= tol, rel_err(s, gs)
        # an affine zero_point is round(-min/scale): a parameter 1 ulp away may land on the
        # other side of a .5 tie (4-bit laplace: -min/scale is 7.5 by construction)
        assert np.abs(z - gz).max() <= 1
    q.enable_quant()
    x0 = dev_tensor(xs[0])
    dq = q(x0).detach().cpu().numpy()
    ref = golden[name + "/dq"]
    if exact:
        assert same_values(dq, ref)
        return
    if kind != "dorefa":
        # the forward itself is exact for the parameters this quantizer holds
        s_fwd, z_fwd = q._qparams_preprocess(x0)
        own, _ = oracle.qdq(xs[0], s_fwd.detach().reshape(-1).cpu().numpy(), z_fwd.detach().reshape(-1).cpu().numpy(),
                            qmin, qmax, ch_axis)
        assert same_values(dq, own)
    if np.array_equal(z, gz):
        # same grid up to the parameter tolerance: within one level of the reference everywhere,
        # identical almost everywhere
        step = np.broadcast_to(np.abs(gs).reshape([-1 if (perch and i == ch_axis) else 1 for i in range(ref.ndim)]), ref.shape)
        assert np.all(np.abs(dq - ref) <= step * 1.001 + 1e-12)
        assert np.mean(np.abs(dq - ref) > step * 1e-3) < 0.02


# --------------------------------------------------------------------------------------
# device-resident calibration driver
# --------------------------------------------------------------------------------------
class _QLinear(torch.nn.Module):
    """Stand-in following the reference's QuantOpr convention (modules/linear.py:30-34)."""

    def __init__(self, lin, wcfg, acfg):
        from sparsebit_amd.common import Backend
        from sparsebit_amd.quantizers import build_quantizer

        super().__init__()
        self.weight, self.bias = lin.weight, lin.bias
        self.weight_quantizer = build_quantizer(wcfg)
        self.input_quantizer = build_quantizer(acfg)
        for q in (self.weight_quantizer, self.input_quantizer):
            q.set_backend(Backend.VIRTUAL)

    def forward(self, x):
        return torch.nn.functional.linear(self.input_quantizer(x), self.weight_quantizer(self.weight), self.bias)


@pytest.mark.parametrize("aobs", ["MINMAX", "PERCENTILE", "MSE"])
def test_device_calibrator_matches_oracle(oracle, aobs):
    from sparsebit_amd.calibration import DeviceCalibrator
    from sparsebit_amd.config import quantizer_config

    torch.manual_seed(3)
    wcfg = lambda: quantizer_config("per-channel-symmetric", 8)
    acfg = lambda: quantizer_config("per-tensor-affine", 8, observer=aobs, target="feature", layout="NLC", alpha=0.01)
    model = torch.nn.Sequential(_QLinear(torch.nn.Linear(48, 64), wcfg(), acfg()), torch.nn.GELU(),
                                _QLinear(torch.nn.Linear(64, 32), wcfg(), acfg())).cuda()
    batches = [torch.randn(3, 17, 48, device="cuda") for _ in range(4)]
    # what each operator sees in a float forward pass
    seen = {0: [], 2: []}
    hooks = [model[i].register_forward_pre_hook(lambda m, a, i=i: seen[i].append(a[0].detach().cpu().numpy())) for i in (0, 2)]
    with torch.no_grad():
        for b in batches:
            model(b)
    for h in hooks:
        h.remove()
    res = DeviceCalibrator(model).calibrate(batches)
    assert len(res) == 4
    # sharded=True in a single process: same protocol, collectives are no-ops -> same result
    res2 = DeviceCalibrator(model).calibrate(batches, sharded=True)
    for k in res:
        assert torch.equal(res[k][0], res2[k][0]) and torch.equal(res[k][1], res2[k][1])
    for i in (0, 2):
        data = np.concatenate([a.reshape(-1) for a in seen[i]])
        if aobs == "MINMAX":
            mn, mx = oracle.minmax(data, 0, False)
            assert len(model[i].input_quantizer.observer.data_cache) == 0  # streamed, nothing cached
        elif aobs == "PERCENTILE":
            mn, mx = oracle.percentile(data, 0.01, per_channel=False)
        if aobs == "MSE":
            s, z, _, _ = oracle.mse(data, 0, 255, False, per_channel=False)
        else:
            s, z = oracle.qparams_from_minmax(mn, mx, 0, 255, False)
        iq = model[i].input_quantizer
        assert np.array_equal(iq.scale.reshape(-1).cpu().numpy(), s) and same_values(iq.zero_point.reshape(-1).cpu().numpy(), z)
        assert list(iq.scale.shape) == [1, 1, 1]
        w = model[i].weight.detach().cpu().numpy()
        ws, wz = oracle.qparams_from_minmax(*oracle.minmax(w, 0, True), -128, 127, True)
        assert np.array_equal(model[i].weight_quantizer.scale.reshape(-1).cpu().numpy(), ws)
    # quantized forward afterwards == oracle QDQ chain on the first operator
    for m in (model[0], model[2]):
        m.input_quantizer.enable_quant()
        m.weight_quantizer.enable_quant()
    x = batches[0]
    iq, wq = model[0].input_quantizer, model[0].weight_quantizer
    xq, _ = oracle.qdq(x.cpu().numpy(), iq.scale.reshape(-1).cpu().numpy(), iq.zero_point.reshape(-1).cpu().numpy(), 0, 255, 2)
    assert same_values(iq(x).cpu().numpy(), xq)


# --------------------------------------------------------------------------------------
# BASELINE.json configs as workload shapes (synthetic tensors; datasets / checkpoints are not
# available offline): each runs the product API at the config's real tensor sizes and is
# checked by oracle-on-a-sample plus size-independent properties.
# --------------------------------------------------------------------------------------
def _sample_rows_check(oracle, x, scale, zp, qmin, qmax, y, rows):
    ref, _ = oracle.qdq(x[rows].float().cpu().numpy(), scale.reshape(-1)[rows].cpu().numpy(),
                        zp.reshape(-1)[rows].cpu().numpy(), qmin, qmax, 0)
    assert same_values(y[rows].float().cpu().numpy(), ref.reshape(y[rows].shape))


def test_config2_resnet50_perchannel_mse_shapes(oracle):
    """ResNet-50 PTQ per-channel 8w8a, MSE observer: the largest 3x3 conv weight (512x512x3x3)
    and a per-tensor MSE activation of 32x256x56x56."""
    from sparsebit_amd.common import Backend
    from sparsebit_amd.config import quantizer_config
    from sparsebit_amd.quantizers import build_quantizer

    g = torch.Generator().manual_seed(50)
    w = (torch.randn(512, 512, 3, 3, generator=g) * 0.05).cuda()
    q = build_quantizer(quantizer_config("per-channel-symmetric", 8, observer="MSE"))
    q.set_backend(Backend.VIRTUAL)
    q.update_observer(w)
    scale, zp = q.calc_qparams()
    rows = [0, 1, 77, 255, 511]
    rs, rz, rbest, _ = oracle.mse(w[rows].cpu().numpy(), -128, 127, True, 0, True)
    assert np.array_equal(q.observer.best_index.cpu().numpy()[rows], rbest)
    assert np.array_equal(scale.reshape(-1)[rows].cpu().numpy(), rs)
    q.enable_quant()
    y = q(w)
    _sample_rows_check(oracle, w.reshape(512, -1), scale, zp, -128, 127, y.reshape(512, -1), rows)
    a = torch.relu(torch.randn(32, 256, 56, 56, generator=g)).cuda()
    qa = build_quantizer(quantizer_config("per-tensor-affine", 8, observer="MSE", target="feature"))
    qa.set_backend(Backend.VIRTUAL)
    for chunk in a.chunk(4):
        qa.update_observer(chunk)
    sa, za = qa.calc_qparams()
    assert 0 <= int(qa.observer.best_index.item()) < 80 and float(sa) > 0
    qa.enable_quant()
    ya = qa(a)
    assert torch.equal(qa(ya), ya)  # idempotent
    assert int(torch.unique(ya).numel()) <= 256


def test_config3_deit_percentile_shapes(oracle):
    """DeiT-small PTQ, percentile observer: four calibration batches of 64x197x384 tokens per
    tensor (the shardable radix path) and a 1536x384 weight per channel (row path)."""
    from sparsebit_amd.config import quantizer_config
    from sparsebit_amd.observers import build_observer
    from sparsebit_amd.quantizers.quant_descriptor import QuantDescriptor

    g = torch.Generator().manual_seed(33)
    xs = [torch.randn(64, 197, 384, generator=g) for _ in range(4)]
    cfg = quantizer_config("per-tensor-symmetric", 8, observer="PERCENTILE", target="feature", layout="NLC", alpha=1e-3)
    obs = build_observer(cfg, QuantDescriptor(cfg))
    for x in xs:
        obs.data_cache.update(x.cuda())
    mn, mx = obs.calc_minmax()
    rmn, rmx = oracle.percentile(np.concatenate([x.numpy().reshape(-1) for x in xs]), 1e-3, per_channel=False)
    assert float(mn) == float(rmn[0]) and float(mx) == float(rmx[0])
    w = torch.randn(1536, 384, generator=g)
    cfgw = quantizer_config("per-channel-symmetric", 8, observer="PERCENTILE", alpha=1e-3)
    obw = build_observer(cfgw, QuantDescriptor(cfgw))
    obw.data_cache.update(w.cuda())
    mn, mx = obw.calc_minmax()
    rmn, rmx = oracle.percentile(w.numpy(), 1e-3, 0, True)
    assert np.array_equal(mn.cpu().numpy(), rmn) and np.array_equal(mx.cpu().numpy(), rmx)


def test_config5_resnet50_qat_lsq_mask_shapes(oracle):
    """ResNet-50 QAT 4w4a LSQ + 50 % unstructured mask: fused mask+QDQ forward == mask multiply
    then quantizer, on the 512x512x3x3 weight; backward through the unfused graph."""
    from sparsebit_amd.common import Backend
    from sparsebit_amd.config import quantizer_config, sparser_config
    from sparsebit_amd.quantizers import build_quantizer
    from sparsebit_amd.sparsers import build_sparser

    g = torch.Generator().manual_seed(5)
    w = (torch.randn(512, 512, 3, 3, generator=g) * 0.03).cuda().requires_grad_(True)
    sp = build_sparser(sparser_config(0.5))
    mask = sp.calc_mask(w)
    assert abs(float(mask.float().mean()) - 0.5) < 1e-3
    rm, rt = oracle.l1_mask(w.detach().cpu().numpy(), 0.5)
    assert float(sp.calc_threshold(w)) == float(rt) and np.array_equal(mask.cpu().numpy(), rm)
    q = build_quantizer(quantizer_config("per-channel-symmetric", 4, quantizer="lsq"))
    q.set_backend(Backend.VIRTUAL)
    q.update_observer(w)
    q.calc_qparams()
    q.enable_quant()
    unfused = q(w * mask)
    fused = q.forward_masked(w.detach(), mask=mask)
    fused_t = q.forward_masked(w.detach(), thresh=sp.calc_threshold(w))
    assert torch.equal(unfused.detach(), fused) and torch.equal(fused, fused_t)
    assert int(fused.reshape(512, -1)[3].unique().numel()) <= 16
    unfused.sum().backward()
    assert w.grad is not None and q.scale.grad is not None
    assert bool((w.grad[~mask] == 0).all())  # pruned weights get no gradient through `w * mask`


def test_minmax_wire_format_kernels(ops):
    """pack -> (MAX over ranks) -> unpack keeps values and NaNs; two simulated ranks."""
    a_mn = torch.tensor([-1.0, float("nan"), 0.5, -3.0], device="cuda")
    a_mx = torch.tensor([2.0, 1.0, float("nan"), 4.0], device="cuda")
    b_mn = torch.tensor([-2.0, 0.0, 0.25, float("-inf")], device="cuda")
    b_mx = torch.tensor([1.0, 5.0, 7.0, float("inf")], device="cuda")
    buf = torch.maximum(ops.minmax_pack(a_mn, a_mx), ops.minmax_pack(b_mn, b_mx))
    mn, mx = ops.minmax_unpack(buf, a_mn.shape)
    assert same_values(mn.cpu().numpy(), np.array([-2.0, np.nan, 0.25, -np.inf], np.float32))
    assert same_values(mx.cpu().numpy(), np.array([2.0, 5.0, np.nan, np.inf], np.float32))
