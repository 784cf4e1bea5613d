"""Run the reference's OWN GPTQ kernel test file -- large_language_models/llama/quantization/test_cuda_kernel.py, unmodified --
on the MI355X, with the one change INTEGRATION.md section 3 describes: `utils/load_cuda_kernel.py` (which JIT-builds the
reference's CUDA extension) answered by `sparsebit_amd.gptq.cuda_kernel`, the stand-in for its pybind module
(cuda/cuda_kernel.cpp:65-73: vecquant{2,3,4}matmul / vecgroupquant{2,3,4}matmul).

    python tools/run_reference_gptq_tests_on_device.py [--reference /path/to/Sparsebit] > profiles/r06_reference_gptq_tests_on_device.log

Everything else is the reference's: `Quantizer.find_params`, `quantize`, `QuantLinear.pack` (its numpy loop), `QuantLinear.forward`
-> `Quant{2,3,4}Matmul.apply` -> the six entry points, and the test's own criterion `assert_allclose(sim_out, gt_out, rtol=1e-5,
atol=1e-5)` against `nn.Linear` on the dequantized weight (test_cuda_kernel.py:21-47).  The reference tree is not part of this
repository (tools/run_reference_on_device.py: same conventions).  `torch.testing.assert_allclose` (deprecated; gone in
newer torch, where `assert_close` with the same rtol / atol answers) is wrapped only to record the largest |difference|.
"""
import argparse
import os
import sys
import time
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default=None)
    ap.add_argument("--only", default=None, help="substring of the test names to run")
    args = ap.parse_args()
    import run_reference_on_device as R

    ref = R.find_reference(args.reference)
    qdir = os.path.join(ref, "large_language_models", "llama", "quantization")
    if not os.path.isfile(os.path.join(qdir, "test_cuda_kernel.py")):
        raise SystemExit("no large_language_models/llama/quantization/test_cuda_kernel.py under %s" % ref)
    import torch

    from sparsebit_amd import gptq

    stub = types.ModuleType("utils.load_cuda_kernel")
    stub.cuda_kernel = gptq.cuda_kernel
    sys.modules["utils.load_cuda_kernel"] = stub  # (utils/quant.py:5 `from .load_cuda_kernel import cuda_kernel`)
    sys.path.insert(0, qdir)
    worst = {"err": 0.0}
    import warnings

    warnings.filterwarnings("ignore", category=FutureWarning)
    inner = getattr(torch.testing, "assert_allclose", None)  # (deprecated since 1.12; gone in newer releases)

    def assert_allclose(actual, expected, rtol, atol):  # the test's criterion, with the largest difference recorded
        worst["err"] = max(worst["err"], float((actual - expected).abs().max()))
        if inner is not None:
            inner(actual, expected, rtol=rtol, atol=atol)
        else:
            torch.testing.assert_close(actual, expected, rtol=rtol, atol=atol)

    torch.testing.assert_allclose = assert_allclose
    import test_cuda_kernel as T

    print("reference test file: %s" % os.path.join(qdir, "test_cuda_kernel.py"))
    print("cuda_kernel: %r" % type(gptq.cuda_kernel))
    names = [n for n in dir(T) if n.startswith("test_") and callable(getattr(T, n))]
    # (file order)
    src = open(os.path.join(qdir, "test_cuda_kernel.py")).read()
    names.sort(key=lambda n: src.index("def " + n))
    failed = 0
    t_all = time.perf_counter()
    for n in names:
        if args.only and args.only not in n:
            continue
        worst["err"] = 0.0
        torch.manual_seed(0)
        t0 = time.perf_counter()
        try:
            getattr(T, n)()
            res = "passed"
        except AssertionError as e:
            res = "FAILED: " + str(e).splitlines()[0][:200]
            failed += 1
        print("%-48s %s   (%.1f s, largest |sim - gt| %.3e)" % (n, res, time.perf_counter() - t0, worst["err"]), flush=True)
    print("%d test functions, %d failed, %.0f s" % (len(names), failed, time.perf_counter() - t_all))
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
