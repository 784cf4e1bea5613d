"""The hot kernels of libsbq.so do not spill vector registers (read from the code objects' metadata; no GPU needed).

A spilled register is reloaded through the same in-order vector-memory path as the kernel's data loads: in round 6 the
fp32 selection kernels sat at the 128-register cap of their 1024-thread workgroups with four slab buffers and spilled 13-38
registers -- 10 us of the model-wide selection's 62 (`group_kth_kernel`).  This test keeps the
kernels that the bench legs time at zero; `python tools/kernel_resources.py --spills` lists the ones that do spill (cold
paths of the percentile kernels, the fp32-output resident forms, 3- / 2-bit strip tiles)."""
import os
import re
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

HOT = [
    r"qdq_resident_kernel<BF16, BF16, [012], 16>",          # the headline launch and its masked forms
    r"qdq_observe_kernel<BF16, BF16, 8>",
    r"group_kth_kernel<F32, 1024, true, [01]>",              # model-wide thresholds / one fp32 tensor's k-th value
    r"win_one_kernel<F32, 1, false, 1024, OneShard>",
    r"win_one_kernel<BF16, 1, false, 1024, OneShard>",
    r"h16_select_kernel<(BF16|F16), [12], (true|false)>",
    r"gptq_mfma_kernel<[12], [234]>",
    r"gptq_strip_kernel<4, 1, 64, 64, true, false, true, (true|false)>",
    r"gptq_strip_kernel<4, 1, 32, 64, true, true, true, false>",
    r"calib_mse_lanes_kernel<F32>",
    r"mse_partial_kernel<F32, true>",
    r"mask_pack_kernel<BF16>",
    r"stats_minmax_kernel<.*>",
]


@pytest.fixture(scope="module")
def table():
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf") or shutil.which("c++filt") is None:
        pytest.skip("llvm-readelf / c++filt not available")
    lib = os.path.join(ROOT, "sparsebit_amd", "libsbq.so")
    if not os.path.exists(lib):
        pytest.skip("libsbq.so not built")
    import kernel_resources

    return kernel_resources.kernels(lib)


def test_library_metadata_is_readable(table):
    assert len(table) > 500
    assert all("vgpr_count" in k and "vgpr_spill_count" in k for k in table)


@pytest.mark.parametrize("pattern", HOT)
def test_hot_kernels_do_not_spill(table, pattern):
    hits = [k for k in table if re.fullmatch(pattern, k["name"])]
    assert hits, "no kernel matches %r" % pattern
    bad = [(k["name"], k["vgpr_spill_count"]) for k in hits if k["vgpr_spill_count"] != 0]
    assert not bad, "vector registers spilled to scratch: %r" % bad
