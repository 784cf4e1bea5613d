#!/bin/bash
# Which kernels a quantized ResNet-20 forward consists of (examples/resnet20_quantopr.py under rocprofv3 --kernel-trace
# --stats): this library's quantizer kernels vs torch's / MIOpen's own.  -> gpurun_out/<tag>_e2e_kernels.txt
TAG=${1:-r05e2e}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o ${TAG} -- python $REPO/examples/resnet20_quantopr.py > $OUT/${TAG}_trace.log 2>&1
find $OUT/${TAG}_trace -name "*kernel_trace.csv" -delete
python - <<PY > $OUT/${TAG}_e2e_kernels.txt
import csv, glob
rows = []
for f in glob.glob("$OUT/${TAG}_trace/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
ours = sum(float(r["TotalDurationNs"]) for r in rows if "sbq" in r["Name"])
calls = sum(int(r["Calls"]) for r in rows)
ours_calls = sum(int(r["Calls"]) for r in rows if "sbq" in r["Name"])
print("# examples/resnet20_quantopr.py (eager, planned, captured and float forwards of a quantized ResNet-20, batch 16) under")
print("# rocprofv3 --kernel-trace --stats: GPU time by kernel.  sbq kernels: %.1f %% of the GPU time (%d of %d dispatches)" % (100 * ours / tot, ours_calls, calls))
print("# share_of_gpu_time  calls  avg_ns  name")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print("%6.2f %%  %7s  %9.0f  %s" % (100 * float(r["TotalDurationNs"]) / tot, r["Calls"], float(r["AverageNs"]), r["Name"][:150]))
PY
rm -rf $OUT/${TAG}_trace
cat $OUT/${TAG}_e2e_kernels.txt
