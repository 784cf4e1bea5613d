#!/bin/bash
# Collect the rocprofv3 kernel-trace summary (and, in separate passes, the HBM PMC counters)
# of the default bench.py command on the GPU box.  Usage: tools/rocprof_bench.sh <tag>
# Writes gpurun_out/<tag>_*.{csv,txt}; copy what should be judged into profiles/.
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --no-cpu-baseline --steps 500 --warmup 50"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o ${TAG} -- $CMD > $OUT/${TAG}_trace.log 2>&1
# PMC passes: FETCH_SIZE and WRITE_SIZE do not fit one pass (TCC has 4 slots: 3 + 2)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -o ${TAG} -- $CMD > $OUT/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -o ${TAG} -- $CMD > $OUT/${TAG}_pmc_write.log 2>&1
# vector-ALU counters (their own pass): the MSE observer is the one VALU-bound kernel of the path
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/${TAG}_pmc_valu -o ${TAG} -- $CMD > $OUT/${TAG}_pmc_valu.log 2>&1
find $OUT -name "${TAG}*stats*.csv" -o -name "${TAG}*counter*.csv" | head
