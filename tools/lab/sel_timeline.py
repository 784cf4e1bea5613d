"""Dev tool: timeline of one call of each windowed-selection entry point from a rocprofv3 kernel trace
(gpurun_out/<tag>_sel_trace, written by tools/rocprof_select.sh): start offsets and durations of its launches."""
import csv, glob, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02v4"
rows = list(csv.DictReader(open(glob.glob("gpurun_out/%s_sel_trace/**/*kernel_trace.csv" % tag, recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r): return r["Kernel_Name"].replace("void sbq::(anonymous namespace)::", "").split("(")[0][:52]
plans = [i for i, r in enumerate(rows) if "win_plan_kernel" in r["Kernel_Name"]]
seen = set()
for n, i0 in enumerate(plans):
    i1 = plans[n + 1] if n + 1 < len(plans) else len(rows)
    seq = rows[i0:i1]
    key = tuple(nm(r) for r in seq[:3])
    # the 6th call of each kind (warm)
    cnt = sum(1 for k in seen if k[0] == key)
    seen.add((key, n))
    if cnt != 5: continue
    t0 = int(seq[0]["Start_Timestamp"])
    for r in seq:
        if "win_" not in r["Kernel_Name"]: break
        print("%-54s start %7.1f us  dur %6.2f us  wgs %4d" % (nm(r), (int(r["Start_Timestamp"]) - t0) / 1e3,
              (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])))
    print()
