#!/bin/bash
# rocprofv3 kernel trace of tools/r06_group_probe.py (the model-wide L1 thresholds), split by variant.
# Usage: tools/r06_group_prof.sh <tag> [pmc]; writes gpurun_out/<tag>_group_*.  Counter passes are separate runs
# (one counter each) and everything is under `timeout`: a rocprofv3 that aborts can hang for its whole lease.
set -u
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/tools/r06_group_probe.py"
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_group_trace -o ${TAG} -- $CMD > $OUT/${TAG}_group_trace.log 2>&1
python - <<PY
import csv, glob, statistics as st
rows = [r for r in csv.DictReader(open(glob.glob("$OUT/${TAG}_group_trace/**/*kernel_trace.csv", recursive=True)[0])) if "group_kth" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
half = len(d) // 2
with open("$OUT/${TAG}_group_kernels.txt", "w") as o:
    for name, part in (("candidate segments", d[:half]), ("round 5 (knob 2 = 34)", d[half:])):
        a, b = part[0::2], part[1::2]
        o.write("%-24s launch 1: median %.2f us (n=%d)   launch 2: median %.2f us   sum %.2f us\n" % (name, st.median(a), len(a), st.median(b), st.median(a) + st.median(b)))
print(open("$OUT/${TAG}_group_kernels.txt").read())
PY
if [ "${2:-}" = "pmc" ]; then
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/${TAG}_group_pmc_$C -o ${TAG} -- $CMD > $OUT/${TAG}_group_pmc_$C.log 2>&1
  done
  python - <<PY
import csv, glob, statistics as st
with open("$OUT/${TAG}_group_pmc.txt", "w") as o:
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob("$OUT/${TAG}_group_pmc_%s/**/*counter_collection.csv" % c, recursive=True)
        if not f:
            o.write("%s: no output\n" % c); continue
        rows = [r for r in csv.DictReader(open(f[0])) if "group_kth" in r["Kernel_Name"] and r["Counter_Name"] == c]
        rows.sort(key=lambda r: int(r.get("Start_Timestamp", r.get("Dispatch_Id", 0))))
        v = [float(r["Counter_Value"]) for r in rows]
        half = len(v) // 2
        for name, part in (("candidate segments", v[:half]), ("round 5 (knob 2 = 34)", v[half:])):
            a, b = part[0::2], part[1::2]
            o.write("%s %-24s launch 1: %.0f   launch 2: %.0f   (raw counter units per launch, median)\n" % (c, name, st.median(a), st.median(b)))
print(open("$OUT/${TAG}_group_pmc.txt").read())
PY
fi
