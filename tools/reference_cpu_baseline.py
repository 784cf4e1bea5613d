"""Time the REAL reference's CPU Quantizer on this host's cores (cpu_baseline.kind = "reference" of bench.py).

    python tools/reference_cpu_baseline.py --reference /path/to/Sparsebit [--seed 0] [--budget-s 20]

`--reference` is a checkout of megvii-research/Sparsebit (the directory that holds sparsebit/); it is never copied and
never part of this repository -- on a box without one, bench.py falls back to oracle/torch_port.py (kind "port").
Runs with the GPUs hidden (the reference's GPU branch would JIT-build its CUDA extension, quant_tensor.py:7-22), with
the three import stubs of tests/golden/gen_golden.py (yacs, onnx, torchvision.ops.stochastic_depth), and executes the
north star's wording literally:

    q = build_quantizer(cfg)                         sparsebit/quantization/quantizers/__init__.py:18-23
    q.update_observer(w); q.calc_qparams()           quantizers/base.py:33-39,66-68 (min-max observer, per channel)
    q.enable_quant(); y = q(w)                       quantizers/base.py:55-64 -> uniform.py:14-16 -> STE -> ort_fake_quant
                                                     (quant_tensor.py:159-185, CPU branch)

on the headline weight (randn * logspace(-2, 1) rows, 4096 x 4096, bf16 values fed as fp32 --
the reference's own fp16 work-around, quant_tensor.py:165-166).  Prints ONE JSON line: elements/s of the forward
(best of 5 per thread count over {all, 64, 32, 16} cores, bounded by --budget-s), the calibration time, and a digest
of the output (bf16 bits, crc32) that bench.py compares with the GPU result of the same weight.
"""
import argparse
import json
import os
import sys
import time
import zlib

os.environ["HIP_VISIBLE_DEVICES"] = ""
os.environ["CUDA_VISIBLE_DEVICES"] = ""

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", required=True)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--budget-s", type=float, default=20.0)
    ap.add_argument("--rows", type=int, default=4096)
    ap.add_argument("--cols", type=int, default=4096)
    args = ap.parse_args()
    if not os.path.isdir(os.path.join(args.reference, "sparsebit")):
        raise SystemExit("no sparsebit/ under %s" % args.reference)
    import torch

    assert not torch.cuda.is_available()
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import gen_golden

    gen_golden.install_stubs()
    sys.path.insert(0, os.path.abspath(args.reference))
    from sparsebit.quantization.common import Backend
    from sparsebit.quantization.quantizers import build_quantizer

    g = torch.Generator().manual_seed(args.seed)
    w = torch.randn(args.rows, args.cols, generator=g) * torch.logspace(-2, 1, args.rows).unsqueeze(1)
    w = w.bfloat16().float()
    cfg = gen_golden.qcfg("per-channel-symmetric", 8, observer="MINMAX", quantizer="uniform", target_weight=True)
    q = build_quantizer(cfg)
    q.set_backend(Backend.VIRTUAL)
    ncpu = os.cpu_count() or 1
    t = time.perf_counter()
    q.update_observer(w)
    scale, zp = q.calc_qparams()
    calib_ms = (time.perf_counter() - t) * 1e3
    q.enable_quant()
    best, best_threads, reps = float("inf"), ncpu, 0
    t_begin = time.perf_counter()
    with torch.no_grad():
        for threads in sorted({ncpu, min(ncpu, 64), min(ncpu, 32), min(ncpu, 16)}, reverse=True):
            torch.set_num_threads(threads)
            y = q(w)  # warm-up
            for _ in range(5):
                a = time.perf_counter()
                y = q(w)
                dt = time.perf_counter() - a
                reps += 1
                if dt < best:
                    best, best_threads = dt, threads
            if time.perf_counter() - t_begin > args.budget_s:
                break
    bits = y.bfloat16().view(torch.int16).numpy().tobytes()
    print(json.dumps({
        "value": round(w.numel() / best, 1),
        "unit": "elements/s",
        "cores": best_threads,
        "host_cores": ncpu,
        "kind": "reference",
        "forward_ms": round(best * 1e3, 3),
        "calibration_ms": round(calib_ms, 2),
        "runs": reps,
        "quantizer": "%s / observer %s" % (type(q).__module__, type(q.observer).__module__),
        "out_bf16_crc32": zlib.crc32(bits) & 0xFFFFFFFF,
        "scale_crc32": zlib.crc32(scale.reshape(-1).float().numpy().tobytes()) & 0xFFFFFFFF,
        "torch_threads_tried": "all / 64 / 32 / 16 of %d" % ncpu,
    }), flush=True)


if __name__ == "__main__":
    main()
