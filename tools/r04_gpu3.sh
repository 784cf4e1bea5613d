#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-r04d}
( timeout 600 python -m pytest tests/test_gpu_r04.py -q -x -k "h16" 2>&1 | tail -15 ) > gpurun_out/${T}_h16tests.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_select_win.py tests/test_gpu_r03.py -q -x 2>&1 | tail -8 ) > gpurun_out/${T}_seltests.log 2>&1
( timeout 200 python tools/lab/h16_stamps.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/${T}_stamps.log 2>&1
( timeout 300 python tools/lab/h16_time.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/${T}_h16_time.log 2>&1
tail -8 gpurun_out/${T}_h16tests.log
tail -4 gpurun_out/${T}_seltests.log
cat gpurun_out/${T}_stamps.log
cat gpurun_out/${T}_h16_time.log
