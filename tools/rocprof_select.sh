#!/bin/bash
# rocprofv3 of the order-statistics entry points (tools/select_bench.py): kernel-trace summary and, in its own
# pass, FETCH_SIZE per kernel.  Usage: tools/rocprof_select.sh <tag>; writes gpurun_out/<tag>_select_*.csv
set -u
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/tools/select_bench.py --quick"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_sel_trace -o ${TAG} -- $CMD > $OUT/${TAG}_sel_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_sel_fetch -o ${TAG} -- $CMD > $OUT/${TAG}_sel_fetch.log 2>&1
python $REPO/tools/select_prof_summary.py $OUT $TAG
