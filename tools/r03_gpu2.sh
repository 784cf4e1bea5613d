#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 200 python tools/lab/sel_debug.py 2>&1 | grep -v amdgpu.ids | tail -22 ) > gpurun_out/r03b_seldebug.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_r03.py tests/test_gpu_select_win.py -x -q 2>&1 | tail -25 ) > gpurun_out/r03b_tests.log 2>&1
( timeout 300 python tools/r03_probe.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r03b_probe.log 2>&1
cat gpurun_out/r03b_seldebug.log; tail -6 gpurun_out/r03b_tests.log; cat gpurun_out/r03b_probe.log
