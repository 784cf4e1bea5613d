#!/bin/bash
# rocprofv3 passes (kernel trace; FETCH_SIZE; WRITE_SIZE -- separate --pmc passes) of the GPTQ decode shapes, summarised
# on the box into gpurun_out/<tag>_gptq_pmc.txt
TAG=${1:-r04g}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/tools/lab/gptq_decode_prof.py"
rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_gtrace -o ${TAG} -- $CMD > $OUT/${TAG}_gtrace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_gfetch -o ${TAG} -- $CMD > $OUT/${TAG}_gfetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_gwrite -o ${TAG} -- $CMD > $OUT/${TAG}_gwrite.log 2>&1
python - <<PY > $OUT/${TAG}_gptq_pmc.txt
import csv, glob, collections
shapes = {(4096, 4096): 9.5e6, (4096, 11008): 25.5e6, (11008, 4096): 25.4e6}
def bytes_of(in_f, out_f):
    return in_f // 8 * out_f * 4 + 2 * out_f * (in_f // 128) * 4 + (in_f + 2 * out_f) * 4
rows = collections.defaultdict(dict)
for f in glob.glob("$OUT/${TAG}_gtrace/**/*kernel_trace.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gptq_strip" not in r["Kernel_Name"]: continue
        acc[(r["Kernel_Name"].split("(")[0][-60:], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"), r.get("Workgroup_Size_X") or r.get("Workgroup_Size"))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in acc.items():
        v = sorted(v)[5:]  # drop the first launches' outliers from the top end? no: sort, drop the 5 smallest -> keep simple: median
        rows[k]["n"] = len(v); rows[k]["median_ns"] = v[len(v) // 2]
for which, name in (("gfetch", "FETCH_SIZE"), ("gwrite", "WRITE_SIZE")):
    for f in glob.glob("$OUT/${TAG}_%s/**/*counter_collection.csv" % which, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "gptq_strip" not in r["Kernel_Name"] or r["Counter_Name"] != name: continue
            acc[(r["Kernel_Name"].split("(")[0][-60:], r["Grid_Size"], r["Workgroup_Size"])].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            rows[k][name] = sum(v) / len(v)
print("# gptq_strip_kernel, 4-bit g128 B=1, HBM-cold; rocprofv3 kernel trace (median duration) and two --pmc passes")
print("# read bytes = 2 x FETCH_SIZE x 1024 (gfx950 correction of the guide), written = WRITE_SIZE x 1024")
for k, d in sorted(rows.items(), key=lambda kv: str(kv[0])):
    print(k, d)
PY
cat $OUT/${TAG}_gptq_pmc.txt
rm -rf $OUT/${TAG}_gtrace $OUT/${TAG}_gfetch $OUT/${TAG}_gwrite
