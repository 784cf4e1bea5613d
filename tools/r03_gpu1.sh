#!/bin/bash
# first GPU pass of round 3: new tests, labs, probes, one bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_r03.py tests/test_gpu_select_win.py -x -q 2>&1 | tail -25 ) > gpurun_out/r03a_tests.log 2>&1
( timeout 120 tools/lab/mse_lab ) > gpurun_out/r03a_mse_lab.log 2>&1
( timeout 120 tools/lab/valu_rate ) > gpurun_out/r03a_valu_rate.log 2>&1
( timeout 300 python tools/r03_probe.py ) > gpurun_out/r03a_probe.log 2>&1
( timeout 300 python tools/resident_sweep.py 4096 8192 16384 24576 ) > gpurun_out/r03a_resident.log 2>&1
( timeout 600 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err
tail -5 gpurun_out/r03a_tests.log; cat gpurun_out/r03a_mse_lab.log gpurun_out/r03a_valu_rate.log gpurun_out/r03a_probe.log; tail -c 1500 gpurun_out/r03a_bench.err
