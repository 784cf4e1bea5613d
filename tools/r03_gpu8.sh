#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 200 python tools/lab/sel_stamps.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r03h_stamps.log 2>&1
( timeout 1200 python -m pytest tests/test_gpu_r03.py tests/test_gpu_select_win.py tests/test_gpu_observe_fused.py -x -q 2>&1 | tail -8 ) > gpurun_out/r03h_tests.log 2>&1
( timeout 300 python tools/r03_probe.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r03h_probe.log 2>&1
( timeout 200 tools/lab/mse_lab ) > gpurun_out/r03h_mse_lab.log 2>&1
grep -E "adv|arrival|end  |sweep|plan|passed|failed|windows|==" gpurun_out/r03h_stamps.log; tail -3 gpurun_out/r03h_tests.log; cat gpurun_out/r03h_probe.log gpurun_out/r03h_mse_lab.log
