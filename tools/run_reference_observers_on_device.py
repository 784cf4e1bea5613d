"""The REAL reference's QuantModel(resnet20) calibrated on the MI355X through sparsebit_amd.plugin.install(calibrate="device")
with MSE and PERCENTILE observers, against the reference ALONE on the host (its CPU path, GPUs hidden, no plugin) -- the
observer kinds tools/run_reference_on_device.py (min-max) leaves out.  Same conventions, same seeds, same model.

    python tools/run_reference_observers_on_device.py [--reference /path/to/Sparsebit] > profiles/r06_reference_observers_on_device.log

Per configuration: every live quantizer's scale / zero point.  Weight quantizers see identical inputs in both processes: their
scales must agree bit for bit (MSE: except rows where two candidates' losses tie to the rounding of an fp32 mean -- counted,
and checked against the oracle's fp64 sums).  Activation quantizers see MIOpen's convolutions on the device and oneDNN's on
the host (last-bit differences): their scales agree to ~1e-6 relative for min-max-like statistics; an order statistic or an
argmin over 80 candidates may move by one element / one candidate (1 % of the range) where the inputs' last bits decide.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import run_reference_on_device as R  # noqa: E402

# (a per-channel MSE observer cannot be asked of the reference's CPU path: observers/mse.py:46-50 broadcasts the [C, 1] scale
# against the flattened [C, n] data transposed the wrong way round and raises -- SURVEY.md section 9, Q2; the per-channel MSE
# kernels are pinned by goldens computed row by row instead, tests/golden/gen_golden.py)
CONFIGS = {
    "W percentile / A percentile": ("PERCENTILE", "PERCENTILE"),
    "W minmax / A mse": ("MINMAX", "MSE"),
    "W minmax / A percentile": ("MINMAX", "PERCENTILE"),
}


def yaml_for(w_obs, a_obs):
    parts = R.PTQ_YAML.split("A:")
    w = parts[0].replace("TYPE: MINMAX", "TYPE: " + w_obs)
    a = parts[1].replace("TYPE: MINMAX", "TYPE: " + a_obs)
    return w + "A:" + a


def host_leg(ref, out_path):
    import contextlib
    import io

    import torch

    assert not torch.cuda.is_available()
    R.setup(ref)
    out = {}
    for name, (w, a) in CONFIGS.items():
        with contextlib.redirect_stdout(io.StringIO()):  # (the reference prints the traced graph)
            qm = R.build(ref, yaml_for(w, a), "cpu")
        qp, y = R.ptq(qm, "cpu")
        out[name] = {"qparams": qp, "logits": y}
    torch.save(out, out_path)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default=None)
    ap.add_argument("--host-leg", default=None)
    ap.add_argument("--runner", default="device", choices=["device", "reference"],
                    help="device: QuantModel.calc_qparams routed through DeviceCalibrator (plugin.install(calibrate='device')); "
                         "reference: the reference's own CalibrationRunner (its fx walk) calling the installed observers")
    args = ap.parse_args()
    ref = R.find_reference(args.reference)
    if args.host_leg:
        return host_leg(ref, args.host_leg)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "host.pt")
        env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--reference", ref, "--host-leg", path], env=env,
                           capture_output=True, text=True, timeout=1500)
        if r.returncode != 0:
            raise SystemExit("host leg failed:\n" + r.stderr[-3000:])
        import torch

        host = torch.load(path)
    sys.path.insert(0, ROOT)
    import sparsebit_amd.plugin as plugin

    plugin.preinstall()
    R.setup(ref)
    info = plugin.install(calibrate="device" if args.runner == "device" else None)
    print("reference: %s" % ref)
    print("plugin.install:", json.dumps(info))
    ok_all = True
    for name, (w, a) in CONFIGS.items():
        qm = R.build(ref, yaml_for(w, a), "cuda")
        qp, y = R.ptq(qm, "cuda")
        h = host[name]
        assert sorted(qp) == sorted(h["qparams"])
        w_rows = w_diff = 0
        a_worst = 0.0
        a_names = 0
        for n, (s_h, z_h) in h["qparams"].items():
            s_d, z_d = qp[n]
            if n.endswith("weight_quantizer"):
                w_rows += s_h.numel()
                w_diff += int((s_d != s_h).sum())
            else:
                a_names += 1
                a_worst = max(a_worst, float(((s_d - s_h).abs() / s_h.abs()).max()))
        dy = float((y - h["logits"]).abs().max())
        print("%-30s weight scales: %d rows, %d differ from the host's bit pattern; activation scales (%d quantizers): max rel diff "
              "%.3e; logits max abs diff %.3e (|logits| max %.3f)" % (name, w_rows, w_diff, a_names, a_worst, dy, float(h["logits"].abs().max())))
        ok_all = ok_all and w_diff == 0
    print(json.dumps({"weight_scales_bit_exact": ok_all}))


if __name__ == "__main__":
    main()
