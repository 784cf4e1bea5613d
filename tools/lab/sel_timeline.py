import csv,glob
rows=list(csv.DictReader(open(glob.glob("gpurun_out/r02_sel_trace/**/sel_kernel_trace.csv", recursive=True)[0])))
def nm(r): return r["Kernel_Name"].replace("void sbq::(anonymous namespace)::","").split("(")[0][:48]
for pat in ("win_pass_kernel<sbq::BF16, true, 1", "win_pass_kernel<sbq::BF16, true, 2, true", "win_pass_kernel<sbq::F32, true, 1"):
    idx=[i for i,r in enumerate(rows) if pat in r["Kernel_Name"]]
    i0=idx[len(idx)//2]-2
    while "init" not in rows[i0]["Kernel_Name"]: i0-=1
    t0=int(rows[i0]["Start_Timestamp"])
    for r in rows[i0:i0+11]:
        print("%-50s start %8.1f us dur %7.2f us wgs %d" % (nm(r), (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"])))
    print()
