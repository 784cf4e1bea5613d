"""Stress test of the single-launch K-split fold of the GPTQ mat-vec (csrc/sbq_gptq.hip: the arrival-counter
protocol of gptq_strip_kernel and gptq_partial_kernel).

10^4 back-to-back calls spread over two HIP streams with separate workspaces, while a third stream keeps the chip
busy with forward-QDQ launches, for batch sizes that take each of the single-launch paths (1 and 2: strip kernel,
4: four-row strip kernel, 8 / 29 / 32: the same kernel walking 2 / 8 / 8 four-row tiles).  Every result must be
bit-equal to its quiet-device result (the sum the protocol must reproduce whoever arrives last), agree with the
two-launch path (knob 2 = 9: partial tiles + a separate fold kernel, no cross-workgroup protocol at all; a different
but equally fixed summation order) at the reference's tolerance, and the arrival counters must be zero afterwards.
Reference kernel: cuda_kernel_4bit.cu:36-81 (one launch, any batch, fp32 atomicAdd -- order-dependent there,
deterministic here).
"""
import pytest
import torch

from sparsebit_amd import lib as L
from sparsebit_amd import ops

pytestmark = pytest.mark.gpu


def _layer(in_f, out_f, bits, gs, seed):
    g = torch.Generator().manual_seed(seed)
    rows = (in_f + 31) // 32 * 3 if bits == 3 else in_f * bits // 32
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (rows, out_f), generator=g, dtype=torch.int64).to(torch.int32)
    groups = in_f // gs
    scales = torch.rand(out_f, groups, generator=g) * 0.02 + 0.001
    zeros = scales * torch.randint(0, 2 ** bits, (out_f, groups), generator=g).float()
    return qw.cuda(), scales.cuda(), zeros.cuda()


@pytest.mark.parametrize("bits", [4, 3])
def test_single_launch_fold_under_concurrency(bits):
    in_f, out_f, gs = 4096, 4096, 128
    qw, scales, zeros = _layer(in_f, out_f, bits, gs, 11 + bits)
    batches = [1, 2, 4, 8, 29, 32]
    g = torch.Generator().manual_seed(3)
    xs = {b: torch.randn(b, in_f, generator=g).cuda() for b in batches}
    bias = {b: torch.randn(b, out_f, generator=g).cuda() for b in batches}
    # ground truth: each path once on a quiet device (what the protocol must reproduce under load is THE SAME sum,
    # whoever arrives last), cross-checked against the two-launch path, which has no cross-workgroup protocol at all
    ref = {}
    for b in batches:
        o = bias[b].clone()
        ops.vecquantmatmul(bits, xs[b], qw, o, scales, zeros, gs)
        torch.cuda.synchronize()
        ref[b] = o
    try:
        L.set_tuning(2, 9)
        for b in batches:
            o = bias[b].clone()
            ops.vecquantmatmul(bits, xs[b], qw, o, scales, zeros, gs)
            torch.cuda.synchronize()
            torch.testing.assert_close(o, ref[b], rtol=1e-5, atol=1e-5)
    finally:
        L.set_tuning(2, 0)
    # background load: QDQ launches on a third stream
    w = torch.randn(4096, 4096, device="cuda").bfloat16()
    ws_, wz = torch.full((4096,), 0.05, device="cuda"), torch.zeros(4096, device="cuda")
    s_bg, s_a, s_b = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    iters = 10_000 // (2 * len(batches)) + 1  # >= 10^4 mat-vec calls in total
    outs = {0: [], 1: []}
    bad = 0
    for it in range(iters):
        with torch.cuda.stream(s_bg):
            ops.fake_quant(w, ws_, wz, -128, 127, 0, torch.bfloat16)
        for si, st in enumerate((s_a, s_b)):
            with torch.cuda.stream(st):
                for b in (batches if si == 0 else batches[::-1]):
                    o = bias[b].clone()
                    ops.vecquantmatmul(bits, xs[b], qw, o, scales, zeros, gs)
                    outs[si].append((b, o))
        if (it + 1) % 64 == 0 or it + 1 == iters:  # check in chunks: keeps memory bounded
            torch.cuda.synchronize()
            for si in (0, 1):
                for b, o in outs[si]:
                    bad += int(not torch.equal(o, ref[b]))
                outs[si].clear()
    assert bad == 0, "%d of %d results differ from the quiet-device result" % (bad, iters * 2 * len(batches))
    # every call left its arrival counters at zero (include/sbq.h: workspace contract)
    for key, buf in ops._gptq_workspaces.items():
        counters = buf[:262144].view(torch.int32)  # SBQ_GPTQ_COUNTER_BYTES
        assert int(counters.abs().sum()) == 0, key


def test_large_shapes_single_launch_agrees_with_two_launch():
    """HBM-sized shapes (the reference's KAT sizes, test_cuda_kernel.py:50-126): 12288 x 49152 is a 302 MB
    4-bit weight stream.  B <= 2 with whole multiples of 512 strips takes gptq_stream_kernel (persistent workers,
    no K split), the others gptq_strip_kernel; knob 2 = 9 is the two-launch path both are compared with."""
    # (6272 x 16384: the persistent-worker kernel with a ragged last pass -- two live K lanes of 32 -- and B = 2)
    for bits, in_f, out_f, b in ((4, 12288, 49152, 1), (4, 8192, 32768, 8), (4, 9216, 36864, 32), (4, 6272, 16384, 1),
                                 (4, 6272, 16384, 2), (4, 8192, 32768, 2), (3, 6272, 16384, 1), (2, 6272, 16384, 1),
                                 (3, 8192, 32768, 1), (2, 8192, 32768, 1), (3, 8192, 16384, 2)):
        qw, scales, zeros = _layer(in_f, out_f, bits, 128, in_f % 97)
        g = torch.Generator().manual_seed(b)
        x = torch.randn(b, in_f, generator=g).cuda()
        o1 = torch.zeros(b, out_f, device="cuda")
        ops.vecquantmatmul(bits, x, qw, o1, scales, zeros, 128)
        try:
            L.set_tuning(2, 9)
            o2 = torch.zeros(b, out_f, device="cuda")
            ops.vecquantmatmul(bits, x, qw, o2, scales, zeros, 128)
        finally:
            L.set_tuning(2, 0)
        # strip kernel vs partial kernel: different (each fixed) summation orders
        torch.testing.assert_close(o1, o2, rtol=1e-4, atol=1e-3)
        o3 = torch.zeros(b, out_f, device="cuda")
        ops.vecquantmatmul(bits, x, qw, o3, scales, zeros, 128)
        assert torch.equal(o1, o3), (bits, in_f, out_f, b)  # deterministic
        del qw, scales, zeros, x, o1, o2
        torch.cuda.empty_cache()
