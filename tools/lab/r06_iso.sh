# lab: rocprofv3 kernel durations of the grouped fp32 selection under the lab knobs (0 = product, 34 = round 5,
# 35 = collect but sweep the tensors again, 36 = collect nothing / read the (stale) segments)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for K in 0 34; do
  rm -rf /tmp/iso_trace
  SBQ_KNOB2=$K timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/iso_trace -o iso -- python $R/tools/lab/r06_group_stamps.py > /tmp/iso.log 2>&1
  python - <<PY
import csv, glob
f = glob.glob("/tmp/iso_trace/**/*kernel_trace.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if "group_kth" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print("knob 2 = $K:", ["%.1f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows])
PY
done
