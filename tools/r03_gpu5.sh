#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_r03.py tests/test_gpu_bench_n2.py -x -q 2>&1 | tail -15 ) > gpurun_out/r03e_tests.log 2>&1
( timeout 600 python tools/r03_gptq_probe.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r03e_gptq.log 2>&1
cat gpurun_out/r03e_tests.log | tail -15; cat gpurun_out/r03e_gptq.log
