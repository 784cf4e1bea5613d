#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-r03o}
true
( timeout 900 python -m pytest tests/test_gpu_select_win.py tests/test_gpu_r03.py -x -q 2>&1 | tail -5 ) > gpurun_out/${T}_tests.log 2>&1
( timeout 300 python tools/r03_probe.py 2>&1 | grep -v amdgpu.ids | grep -E "kth|DeiT" ) > gpurun_out/${T}_probe.log 2>&1
( timeout 300 python tools/lab/sel_sweep_params.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/${T}_params.log 2>&1
grep -E "==|plan done|sweep|counters|arrival|advance|end  |adv0" gpurun_out/${T}_stamps.log; tail -3 gpurun_out/${T}_tests.log; cat gpurun_out/${T}_probe.log; cat gpurun_out/${T}_params.log
