#!/bin/bash
# rocprofv3 passes of the bench command on the GPU box; the per-dispatch counter files are tens of MB each, so they are
# summarised THERE (tools/pmc_summary.py) and only the summaries travel back
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-r04v1}
bash tools/rocprof_bench.sh $T 2>&1 | tail -3
python tools/pmc_summary.py $T > gpurun_out/${T}_summary.json 2> gpurun_out/${T}_summary.err
mkdir -p gpurun_out/${T}_profiles
cp profiles/${T}_* profiles/pmc_latest.json gpurun_out/${T}_profiles/ 2>/dev/null
rm -rf gpurun_out/${T}_pmc_fetch gpurun_out/${T}_pmc_write gpurun_out/${T}_pmc_valu gpurun_out/${T}_trace
ls -la gpurun_out/${T}_profiles; tail -3 gpurun_out/${T}_summary.err; head -12 gpurun_out/${T}_summary.json
