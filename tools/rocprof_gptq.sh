#!/bin/bash
# rocprofv3 of the GPTQ 4-bit mat-vec at the reference's KAT sizes (tools/gptq_prof.py): kernel-trace summary and,
# in its own pass, FETCH_SIZE per kernel and grid.  Usage: tools/rocprof_gptq.sh <tag> [knob2]
set -u
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/tools/gptq_prof.py ${2:-0}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_gptq_trace -o ${TAG} -- $CMD > $OUT/${TAG}_gptq_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_gptq_fetch -o ${TAG} -- $CMD > $OUT/${TAG}_gptq_fetch.log 2>&1
python $REPO/tools/select_prof_summary.py $OUT $TAG gptq
