#!/bin/bash
# Collect the rocprofv3 kernel-trace summary (and, in separate passes, the HBM / VALU PMC counters) of the default
# bench.py command on the GPU box.  Usage: tools/rocprof_bench.sh <tag>
# Summaries go to profiles/<tag>_* and profiles/pmc_latest.json on the box (tools/pmc_summary.py) and are copied to
# gpurun_out/<tag>_summary/ so that they travel back; the raw per-dispatch CSVs (hundreds of MB) are deleted -- gpurun
# merges at most 64 MiB.  The e2e leg (torch / MIOpen kernels of a ResNet-20, ~10^5 dispatches) is skipped under the profiler.
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SBQ_BENCH_SKIP_E2E=1
CMD="python $REPO/bench.py --no-cpu-baseline --steps 500 --warmup 50"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o ${TAG} -- $CMD > $OUT/${TAG}_trace.log 2>&1
find $OUT/${TAG}_trace -name "*kernel_trace.csv" -delete
# PMC passes (no tracing with them): FETCH_SIZE and WRITE_SIZE do not fit one pass (TCC has 4 slots: 3 + 2)
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -o ${TAG} -- $CMD > $OUT/${TAG}_pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -o ${TAG} -- $CMD > $OUT/${TAG}_pmc_write.log 2>&1
# vector-ALU counters (their own pass): the MSE observer is the one VALU-bound kernel of the path
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/${TAG}_pmc_valu -o ${TAG} -- $CMD > $OUT/${TAG}_pmc_valu.log 2>&1
cd $REPO && python tools/pmc_summary.py ${TAG} > $OUT/${TAG}_pmc_summary.log 2>&1
mkdir -p $OUT/${TAG}_summary
cp profiles/${TAG}_* profiles/pmc_latest.json $OUT/${TAG}_summary/ 2>/dev/null
find $OUT/${TAG}_trace -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_summary/ \;
rm -rf $OUT/${TAG}_trace $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_pmc_valu
ls -la $OUT/${TAG}_summary; tail -3 $OUT/${TAG}_pmc_summary.log
