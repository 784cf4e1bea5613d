#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_r03.py tests/test_gpu_plugin.py -x -q 2>&1 | tail -12 ) > gpurun_out/r03g_tests.log 2>&1
tail -6 gpurun_out/r03g_tests.log
bash tools/rocprof_bench.sh r03v1 2>&1 | tail -5
ls -la gpurun_out/r03v1_trace gpurun_out/r03v1_pmc_fetch 2>/dev/null | head; tail -3 gpurun_out/r03v1_trace.log
python tools/pmc_summary.py r03v1 2>&1 | tail -30
# keep what comes back small: the raw per-dispatch traces are large
rm -f gpurun_out/r03v1_trace/*kernel_trace.csv
du -sh gpurun_out
