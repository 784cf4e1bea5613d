#!/bin/bash
# round 4, GPU call 2: the full-histogram selection engine -- correctness first, then timing
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-r04b}
( timeout 600 python -m pytest tests/test_gpu_r04.py -q -x -k "h16 or gptq_mse or streaming" 2>&1 | tail -30 ) > gpurun_out/${T}_h16tests.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_select_win.py tests/test_gpu_r03.py tests/test_gpu_fuzz.py -q -x 2>&1 | tail -30 ) > gpurun_out/${T}_seltests.log 2>&1
( timeout 300 python tools/lab/h16_time.py ) > gpurun_out/${T}_h16_time.log 2>&1
tail -30 gpurun_out/${T}_h16tests.log
tail -12 gpurun_out/${T}_seltests.log
cat gpurun_out/${T}_h16_time.log
